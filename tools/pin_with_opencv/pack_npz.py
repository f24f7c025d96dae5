#!/usr/bin/env python3
"""pin_dump's binary stream -> tests/golden/opencv42_v1.npz.   usage: pack_npz.py dump.bin out.npz"""
import struct
import sys

import numpy as np

DT = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64}


def read(path):
    raw = open(path, "rb").read()
    pos, out = 0, {}
    while pos < len(raw):
        (nl,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        name = raw[pos:pos + nl].decode()
        pos += nl
        dt = raw[pos]
        pos += 1
        (nd,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        dims = struct.unpack_from(f"<{nd}I", raw, pos)
        pos += 4 * nd
        n = int(np.prod(dims)) if nd else 1
        a = np.frombuffer(raw, DT[dt], n, pos).reshape(dims).copy()
        pos += a.nbytes
        out[name] = a
    return out


if __name__ == "__main__":
    d = read(sys.argv[1])
    np.savez_compressed(sys.argv[2], **d)
    print(f"{len(d)} arrays, OpenCV {bytes(d['opencv_version']).decode()} -> {sys.argv[2]}")
