"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (VERDICT r5 #8a).  GPU sanitizers do not exist on this
pool; what CAN be sanitized is (1) the product's host-compilable core -- octree, introsort replay, geometry tables, float
helpers: visual_sgraphs_amd/csrc/*.h through tests/_hostcore, the code whose indices decide keypoint ORDER -- and (2) the
oracle itself.  Each test builds the `asan` target and runs the ordinary test file against it in a child interpreter with
libasan preloaded; any report fails the run (-fno-sanitize-recover, halt_on_error)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.sep in p and Path(p).exists() else None


def _run_under_asan(test_file, extra_env, select=None):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("gcc's libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **extra_env)
    cmd = [sys.executable, "-m", "pytest", str(ROOT / "tests" / test_file), "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"]
    if select:
        cmd += ["-k", select]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(ROOT), timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out
    return out


def test_hostcore_is_clean_under_asan_and_ubsan():
    _run_under_asan("test_hostcore.py", {"VSG_HOSTCORE_ASAN": "1"})


def test_oracle_is_clean_under_asan_and_ubsan():
    _run_under_asan("test_oracle.py", {"VSG_ORACLE_LIB": str(ROOT / "oracle" / "liborb_oracle_asan.so")})
