#!/usr/bin/env python3
"""Harness self-test (NOT a pin): writes a dump in pin_dump's binary format from the CPU ORACLE's outputs, packs it and
runs tests/test_pin_opencv42.py against it, then deletes the file again.  It proves that the dump format, pack_npz.py
and the conditional tests fit together in an image that has no OpenCV; it says nothing about OpenCV.  The npz it
creates must never be committed."""
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle_lib as ol  # noqa: E402
from visual_sgraphs_amd import synth  # noqa: E402

CODE = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}
CASES = [("qvga_rgbd", 320, 240, 42, 1, 500, 4, 0, 0), ("qvga_partial_lapping", 320, 240, 42, 1, 300, 3, 100, 220)]


def put(f, name, a):
    a = np.ascontiguousarray(a)
    f.write(struct.pack("<I", len(name)) + name.encode() + bytes([CODE[a.dtype]]) + struct.pack("<I", a.ndim))
    f.write(struct.pack(f"<{a.ndim}I", *a.shape) + a.tobytes())


def selftest_dbow2():
    """The same for the DBoW2 target: an oracle-generated dump in pin_dbow2's format through pack_npz.py and
    tests/test_pin_dbow2.py."""
    target = ROOT / "tests" / "golden" / "dbow2_v1.npz"
    assert not target.exists(), "a real pin exists: not touching it"
    with tempfile.TemporaryDirectory() as td:
        dump = Path(td) / "dbow2.bin"
        with open(dump, "wb") as f:
            put(f, "opencv_version", np.frombuffer(b"ORACLE-FAKE", np.uint8))
            for name, k, L, seed, sc, wt, lu, nd in [("k10_L3_l2", 10, 3, 31, 1, 1, 2, 400), ("k6_L4_bin", 6, 4, 64, 5, 3, 5, 400)]:
                blob = synth.synthetic_vocabulary(k, L, seed=seed, scoring=sc, weighting=wt)
                d = synth.random_descriptors(nd, 100 + seed)
                nn = (len(blob) - 16) // 45
                raw = np.frombuffer(blob, np.uint8)
                for i in range(0, nd, 7):
                    o = 16 + ((37 * i) % nn) * 45 + 5
                    d[i] = raw[o:o + 32]
                v = ol.OracleVocabulary(blob)
                r = v.transform(d, lu)
                put(f, name + "/params", np.array([k, L, seed, sc, wt, lu, nd, v.nwords], np.int32))
                put(f, name + "/desc", d)
                put(f, name + "/bow_ids", r["bow_ids"].astype(np.int32))
                put(f, name + "/bow_vals", r["bow_vals"].astype(np.float64))
                put(f, name + "/fv_node", r["fv"][0].astype(np.int32))
                put(f, name + "/fv_off", r["fv"][1].astype(np.int32))
                put(f, name + "/fv_idx", r["fv"][2].astype(np.int32))
                put(f, name + "/word", r["word"].astype(np.int32))
            a, b = synth.random_descriptors(512, 901), synth.random_descriptors(512, 902)
            a[0], b[0], b[1] = 0, 255, a[1]
            put(f, "forb/a", a)
            put(f, "forb/b", b)
            put(f, "forb/distance", np.unpackbits(a ^ b, axis=1).sum(1).astype(np.int32))
            put(f, "layout/facts", np.array([28, 0, 8, 12, 16, 20, 24, 1, 32, 1, 8, -1], np.int32))
        try:
            subprocess.check_call([sys.executable, str(Path(__file__).parent / "pack_npz.py"), str(dump), str(target)])
            return subprocess.call([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_pin_dbow2.py"), "-q",
                                    "-m", "not gpu"])
        finally:
            target.unlink(missing_ok=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dbow2":
        sys.exit(selftest_dbow2())
    target = ROOT / "tests" / "golden" / "opencv42_v1.npz"
    assert not target.exists(), "a real pin exists: not touching it"
    with tempfile.TemporaryDirectory() as td:
        dump = Path(td) / "dump.bin"
        with open(dump, "wb") as f:
            put(f, "opencv_version", np.frombuffer(b"ORACLE-FAKE", np.uint8))
            for name, w, h, seed, div, nf, nl, l0, l1 in CASES:
                e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
                mono, kps, desc = e(synth.frame(w, h, seed, amplitude_div=div), (l0, l1))
                put(f, name + "/params", np.array([w, h, seed, div, nf, nl, l0, l1], np.int32))
                put(f, name + "/mono", np.array([mono], np.int32))
                put(f, name + "/kps", kps.view(np.uint8).reshape(len(kps), 28))
                put(f, name + "/desc", desc)
                for l in range(nl):
                    put(f, f"{name}/pyr{l}", e.pyramid_level(l, with_border=True))
                    b = e.blurred_level(l)
                    put(f, f"{name}/blur{l}", b if b is not None else ol.gaussian_blur7(e.pyramid_level(l)))
            # pin_dump's content-class cases (kind index = position in synth.CONTENT_CLASSES)
            ccases = [(k, 320, 240, 5000, 1, 500, 4) for k in range(len(synth.SYNTH_CLASSES))]
            ccases += [(synth.CONTENT_CLASSES.index(n), 640, 480, 5000, 3, 1000, 8) for n in ("value_noise", "defocus", "grating")]
            for kind, w, h, seq, t_, nf, nl in ccases:
                e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
                mono, kps, desc = e(synth.content_frame(synth.CONTENT_CLASSES[kind], w, h, seq, t_))
                name = f"content_{kind}_{w}"
                put(f, name + "/content_params", np.array([w, h, seq, t_, nf, nl, kind, 0], np.int32))
                put(f, name + "/mono", np.array([mono], np.int32))
                put(f, name + "/kps", kps.view(np.uint8).reshape(len(kps), 28))
                put(f, name + "/desc", desc)
                for l in range(nl):
                    put(f, f"{name}/pyr{l}", e.pyramid_level(l, with_border=True))
                    b = e.blurred_level(l)
                    put(f, f"{name}/blur{l}", b if b is not None else ol.gaussian_blur7(e.pyramid_level(l)))
            # pin_dump's photo cases (export_photo_frames.py; index = position in synth.PHOTO_CLASSES)
            from export_photo_frames import PHOTO_CASES
            for idx, w, h, seq, t_, nf, nl in PHOTO_CASES:
                e = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
                mono, kps, desc = e(synth.content_frame(synth.PHOTO_CLASSES[idx], w, h, seq, t_))
                name = f"photo_{idx}_{w}"
                put(f, name + "/photo_params", np.array([w, h, seq, t_, nf, nl, idx, 0], np.int32))
                put(f, name + "/mono", np.array([mono], np.int32))
                put(f, name + "/kps", kps.view(np.uint8).reshape(len(kps), 28))
                put(f, name + "/desc", desc)
                for l in range(nl):
                    put(f, f"{name}/pyr{l}", e.pyramid_level(l, with_border=True))
                    b = e.blurred_level(l)
                    put(f, f"{name}/blur{l}", b if b is not None else ol.gaussian_blur7(e.pyramid_level(l)))
            t = [18, 34, 49, 55, 49, 34, 18]
            put(f, "gauss/kernel_f64", np.array(t, np.float64) / 256)
            put(f, "gauss/line_response", np.array([[(257 * k * 255 + 32768) >> 16 for k in t]], np.uint8))
            put(f, "gauss/flat200_response", np.array([[(200 * 257 * 257 + 32768) >> 16] * 7], np.uint8))
            rng = np.random.default_rng(1)
            rgb = rng.integers(0, 256, (4096, 3), dtype=np.uint8)
            put(f, "gray/rgb", rgb)
            put(f, "gray/rgba", np.concatenate([rgb, rgb[:, :1]], 1))
            img3 = rgb.reshape(64, 64, 3)
            put(f, "gray/rgb2gray", ol.cvt_gray(img3, True))
            put(f, "gray/bgr2gray", ol.cvt_gray(img3, False))
            put(f, "gray/rgba2gray", ol.cvt_gray(img3, True))
            put(f, "gray/bgra2gray", ol.cvt_gray(img3, False))
            img = synth.frame(320, 240, 42)
            put(f, "cv/resize_267x200", ol.resize_linear(img, 267, 200))
            for th in (20, 7):
                x, y, r = ol.fast9_16(img, th, True)
                put(f, f"cv/fast{th}", np.stack([x, y, r], 1).astype(np.int32))
            args = np.array([(y * 123.5, x * 77.25) for y in range(-40, 41, 3) for x in range(-40, 41, 3)], np.float32)
            put(f, "cv/fastatan2_args", args)
            put(f, "cv/fastatan2", np.array([ol.fast_atan2(a, b) for a, b in args], np.float32))
            for cam_name, cam in ol.CAMERAS.items():  # pin_dump's undistort/ block from the oracle's restatement
                r = synth.splitmix64(0xD157, 4096)
                pts = np.stack([((r % np.uint64(2720)).astype(np.int64) - 80) * 0.25,
                                (((r >> np.uint64(20)) % np.uint64(2080)).astype(np.int64) - 80) * 0.25], 1).astype(np.float32)
                pts = np.concatenate([pts, np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32)])
                put(f, f"undistort/{cam_name}/K4", np.asarray(cam["K4"], np.float32))
                put(f, f"undistort/{cam_name}/dist", np.asarray(cam["dist"], np.float32))
                put(f, f"undistort/{cam_name}/in", pts)
                put(f, f"undistort/{cam_name}/out", ol.undistort_points(pts, cam))
        try:
            subprocess.check_call([sys.executable, str(Path(__file__).parent / "pack_npz.py"), str(dump), str(target)])
            rc = subprocess.call([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_pin_opencv42.py"), "-q",
                                  "-m", "not gpu"])
        finally:
            target.unlink(missing_ok=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
