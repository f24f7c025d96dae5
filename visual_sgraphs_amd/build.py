"""In-tree build of the HIP C-ABI library (libvsg_orb.so) for gfx950.

`python -m visual_sgraphs_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libvsg_orb.so"
SOURCES = ["vsg_kernels.hip", "vsg_orb.hip", "vsg_match.hip", "vsg_grid.hip", "vsg_bow.hip", "vsg_frame.hip", "vsg_ctx.hip", "vsg_shard.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wno-unused-value",
         # MFMA results in VGPRs (the matcher's epilogue is VALU): no v_accvgpr_read per accumulator register
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


STAMP = CSRC / "_obj" / "linked_flags.txt"  # flag key of the objects libvsg_orb.so was last linked from


def _flag_key(extra):
    return hashlib.sha256(" ".join(FLAGS + ["--"] + list(extra)).encode()).hexdigest()[:12]


def _flag_stamp():
    try:
        return STAMP.read_text().strip()
    except OSError:
        return ""


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (needed to build libvsg_orb.so; there is no CPU fallback)")


def needs_build():
    if not LIB.exists():
        return True
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + \
        [PKG.parent / "include" / "vsg_orb.h"]
    return any(d.stat().st_mtime > LIB.stat().st_mtime for d in deps)


def build(force=False, verbose=False):
    """Per-source objects compiled in parallel (only the sources that changed, or all of them after a header change),
    then one link.  Objects live in csrc/_obj/<hash of the full flag string> (git-ignored): an object compiled with
    VSG_EXTRA_FLAGS (experiment builds, tools/build_variant.sh) can never be linked into a later normal build.  A
    build with extra flags always links (the library on disk may come from other flags)."""
    extra = os.environ.get("VSG_EXTRA_FLAGS", "").split()
    # an up-to-date library is taken as it is; only a stamp that is PRESENT and names other flags (an experiment build
    # was linked last) forces a relink -- a missing stamp (packaged artefact, copied tree: _obj/ is git-ignored) does not
    if not force and not extra and not needs_build() and _flag_stamp() in ("", _flag_key([])):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in FLAGS if f != "-shared"]
    key = _flag_key(extra)
    obj_dir = CSRC / "_obj" / key
    obj_dir.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [PKG.parent / "include" / "vsg_orb.h", Path(__file__)]
    hdr_time = max(h.stat().st_mtime for h in headers)

    def compile_one(src):
        obj = obj_dir / (src + ".o")
        if not force and obj.exists() and obj.stat().st_mtime > max(hdr_time, (CSRC / src).stat().st_mtime):
            return obj
        cmd = [hipcc] + cflags + extra + ["-c", "-o", str(obj), str(CSRC / src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=str(CSRC))
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs] + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=str(CSRC))
    STAMP.write_text(key + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=bool(os.environ.get("VSG_BUILD_VERBOSE"))))
