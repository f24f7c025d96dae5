#!/usr/bin/env python3
"""Writes the photo cases of the pin recipe as a flat file pin_dump reads (its optional second argument): records of
8 x int32 {w, h, seq, t, nfeatures, nlevels, index in synth.PHOTO_CLASSES, 0} followed by w * h gray bytes.  The frames come from
the committed gray planes (tests/golden/photos_v1.npz) through synth.content_frame -- the same frames the GPU tests, the fuzzer
and bench.py's content sweep use.

    python3 tools/pin_with_opencv/export_photo_frames.py /tmp/pin/photo_frames.bin
    /tmp/pin/pin_dump /tmp/pin/dump.bin /tmp/pin/photo_frames.bin
"""
import struct
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from visual_sgraphs_amd import synth  # noqa: E402

# every photograph at the headline geometry; the building photo also at the stereo (C3) and the 720p (C4) geometry
PHOTO_CASES = [(i, 640, 480, 5000, 3, 1000, 8) for i in range(len(synth.PHOTO_CLASSES))]
PHOTO_CASES += [(0, 752, 480, 5000, 3, 1200, 8), (0, 1280, 720, 5000, 3, 2000, 8)]


def main():
    with open(sys.argv[1], "wb") as f:
        for idx, w, h, seq, t, nf, nl in PHOTO_CASES:
            img = synth.content_frame(synth.PHOTO_CLASSES[idx], w, h, seq, t)
            f.write(struct.pack("<8i", w, h, seq, t, nf, nl, idx, 0) + img.tobytes())
    print(f"wrote {sys.argv[1]}: {len(PHOTO_CASES)} frames")


if __name__ == "__main__":
    main()
