"""Makes tests/golden/photos_v1.npz: three real photographs as gray frames (VERDICT r5 "next" #1).

The datasets BASELINE.json's configurations name (TUM fr1/desk, EuRoC MH_01) are not in this image; three photographs are
(scikit-learn's sample images, matplotlib's sample data).  This script decodes them HERE (authoring container, Pillow),
converts them with the repo's own restatement of the reference's gray conversion (oracle `or_cvt_gray_u8`: [OCV 4.2]
`(R 4899 + G 9617 + B 1868 + 8192) >> 14`, `Tracking.cc:1595-1608`) and commits the gray planes as DATA.  Nothing on the GPU
box decodes a JPEG; `visual_sgraphs_amd.synth.content_frame("photo_*", ...)` reads the planes and derives frames of any
geometry from them by integer operations only (mirror tiling, crop, translation, seeded sensor noise).

    python tests/golden/make_photos.py            # writes tests/golden/photos_v1.npz

Attribution (the licences travel with the data, `photos_v1.npz["attribution"]` holds this text too):
"""
import io
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))

ATTRIBUTION = """\
china    scikit-learn sample image `china.jpg` (640x427): photograph by Flickr user danielbuechele,
         https://www.flickr.com/photos/danielbuechele/6061409035/ , CC BY 2.0 (https://creativecommons.org/licenses/by/2.0/);
         converted to 8-bit gray, otherwise unchanged.
flower   scikit-learn sample image `flower.jpg` (640x427): photograph by Flickr user vultilion,
         https://www.flickr.com/photos/vultilion/6056698931/ , CC BY 2.0; converted to 8-bit gray, otherwise unchanged.
hopper   matplotlib sample data `grace_hopper.jpg` (512x600): official U.S. Navy portrait of Grace Hopper, public domain
         (a work of the U.S. federal government); converted to 8-bit gray, otherwise unchanged.
"""
__doc__ += ATTRIBUTION


def _decode(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def main():
    import matplotlib
    import sklearn

    import oracle_lib as ol
    sk = Path(sklearn.__file__).parent / "datasets" / "images"
    mp = Path(matplotlib.__file__).parent / "mpl-data" / "sample_data"
    src = {"china": sk / "china.jpg", "flower": sk / "flower.jpg", "hopper": mp / "grace_hopper.jpg"}
    out = {}
    for name, path in src.items():
        rgb = _decode(path)
        gray = ol.cvt_gray(rgb, rgb_order=True)
        # the same formula in numpy, so that the fixture does not rest on the C restatement alone
        r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
        assert np.array_equal(gray, ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))
        out[name] = gray
        print(f"{name}: {gray.shape[1]}x{gray.shape[0]} mean {gray.mean():.1f}")
    out["attribution"] = np.frombuffer(ATTRIBUTION.encode(), dtype=np.uint8)
    buf = io.BytesIO()
    np.savez_compressed(buf, **out)
    dst = HERE / "photos_v1.npz"
    dst.write_bytes(buf.getvalue())
    print(f"wrote {dst} ({os.path.getsize(dst)} bytes)")


if __name__ == "__main__":
    main()
