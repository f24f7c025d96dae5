#!/usr/bin/env python3
"""Identity of the library SOURCES a profile belongs to: sha256 (first 16 hex digits) over the kernel / host sources
of libvsg_orb.so in a fixed order.  profiles/traffic_rNN.json records it; bench.py refuses PMC figures whose hash is
not the hash of the sources it runs (the built .so is git-ignored, its sources are not)."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def source_hash():
    csrc = ROOT / "visual_sgraphs_amd" / "csrc"
    files = sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.h")) + list(csrc.glob("*.inc"))) + [ROOT / "include" / "vsg_orb.h"]
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode() + b"\0" + f.read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
