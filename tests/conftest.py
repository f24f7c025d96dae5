import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# PyTorch-ROCm bundles its own HIP runtime; libvsg_orb.so links the system one.  Both can live in one process only
# if torch comes first (bench.py imports it first; the one GPU test that uses torch tensors would otherwise find
# "No HIP GPUs" after the C ABI has initialised the system runtime).
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
