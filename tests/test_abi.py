"""CPU tests of the C-ABI boundary: the library builds/loads, exports exactly what include/vsg_orb.h
declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from visual_sgraphs_amd import build, orb
    build.build()
    return orb.load_library()


def test_header_symbols_are_exported(lib):
    from visual_sgraphs_amd import orb
    header = (ROOT / "include" / "vsg_orb.h").read_text()
    declared = sorted(set(re.findall(r"\b(vsg_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vsg_orb.h but not exported"
    assert sorted(orb.EXPORTS) == declared


def test_keypoint_record_is_cv_keypoint_layout():
    from visual_sgraphs_amd import orb
    assert orb.KP_DTYPE.itemsize == 28
    assert [orb.KP_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_no_cpu_fallback_without_device(lib):
    from visual_sgraphs_amd import orb
    if lib.vsg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(orb.VsgError) as ei:
        orb.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code == -4  # VSG_ERR_NO_DEVICE
    m = orb.ORBmatcher(0.7, True)
    with pytest.raises(orb.VsgError):
        m.block_best2(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))


def test_product_does_not_reference_the_oracle():
    """The shipped path must not import, link or call anything under oracle/."""
    pkg = ROOT / "visual_sgraphs_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")) + [ROOT / "include/vsg_orb.h"]:
        txt = f.read_text()
        assert "orb_oracle" not in txt and "oracle_lib" not in txt and "liborb_oracle" not in txt, f
