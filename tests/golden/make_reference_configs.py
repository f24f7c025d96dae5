#!/usr/bin/env python3
"""Collect every distinct (image size, ORBextractor parameters) tuple the reference ships in its settings files
(/root/reference/config/**/*.yaml: Camera.width/height or Camera1.*, ORBextractor.nFeatures / scaleFactor / nLevels /
iniThFAST / minThFAST -- read by Settings::readORB, orb_slam3/src/Settings.cc, and Tracking::ParseORBParamFile,
Tracking.cc) into tests/golden/reference_configs.json.  The fixture is DATA (numbers from the settings files plus the
names of the files that hold them); the parity tests run the HIP extractor against the oracle for every tuple.

Run in the build container only (the reference does not exist on the GPU box):  python tests/golden/make_reference_configs.py
"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/config")
KEYS = {"w": r"Camera1?\.width", "h": r"Camera1?\.height", "nFeatures": r"ORBextractor\.nFeatures",
        "scaleFactor": r"ORBextractor\.scaleFactor", "nLevels": r"ORBextractor\.nLevels",
        "iniThFAST": r"ORBextractor\.iniThFAST", "minThFAST": r"ORBextractor\.minThFAST"}


def parse(path):
    text = path.read_text(errors="replace")
    out = {}
    for k, pat in KEYS.items():
        m = re.search(r"^\s*" + pat + r"\s*:\s*([0-9.]+)", text, re.M)
        if not m:
            return None
        out[k] = float(m.group(1)) if k == "scaleFactor" else int(float(m.group(1)))
    return out


def parse_stereo(path):
    """Stereo rigs: baseline Stereo.b [m] and Camera1.fx [px] (new settings format) or Camera.bf = b * fx and Camera.fx
    (old format): Frame::mb = mbf / fx and Frame::mbf = b * fx feed ComputeStereoMatches (Frame.cc:957-1127)."""
    text = path.read_text(errors="replace")

    def num(pat):
        m = re.search(r"^\s*" + pat + r"\s*:\s*([-0-9.eE+]+)", text, re.M)
        return float(m.group(1)) if m else None

    fx = num(r"Camera1?\.fx")
    b, bf = num(r"Stereo\.b"), num(r"Camera\.bf")
    if fx is None or (b is None and bf is None):
        return None
    return {"b": b if b is not None else bf / fx, "fx": fx}


def main():
    seen, stereo = {}, {}
    for f in sorted(ROOT.rglob("*.yaml")):
        p = parse(f)
        if p is None:
            continue
        key = tuple(p[k] for k in KEYS)
        seen.setdefault(key, {"params": p, "files": []})["files"].append(str(f.relative_to(ROOT)))
        st = parse_stereo(f)
        if st is not None and f.relative_to(ROOT).parts[0].startswith("Stereo"):
            skey = key + (st["b"], st["fx"])
            stereo.setdefault(skey, {"params": dict(p, **st), "files": []})["files"].append(str(f.relative_to(ROOT)))
    cases = [seen[k] for k in sorted(seen)]
    scases = [stereo[k] for k in sorted(stereo)]
    out = Path(__file__).parent / "reference_configs.json"
    out.write_text(json.dumps({"source": "reference config/**/*.yaml", "cases": cases, "stereo_cases": scases},
                              indent=1) + "\n")
    print(f"{len(cases)} distinct extractor configurations from {sum(len(c['files']) for c in cases)} settings files, "
          f"{len(scases)} distinct stereo rigs -> {out}")


if __name__ == "__main__":
    main()
