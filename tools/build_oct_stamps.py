#!/usr/bin/env python3
"""Diagnostic build for tools/oct_stamps.py: a COPY of visual_sgraphs_amd/csrc with s_memtime stamps at the phase boundaries
of DistributeOctTree (tid 0 of every level's workgroup of frame 0; a debug export reads them back), compiled into
tools/_bin/libvsg_octstamp<suffix>.so.  The tree itself is never touched; the real kernel executes no stamp.
    python tools/build_oct_stamps.py [suffix] [-DVSG_...]"""
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "visual_sgraphs_amd" / "csrc"


def sub(s, old, new, what):
    assert old in s, f"anchor not found: {what}"
    return s.replace(old, new, 1)


def main():
    suffix = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    with tempfile.TemporaryDirectory() as td:
        d = Path(td) / "csrc"
        shutil.copytree(CSRC, d, ignore=shutil.ignore_patterns("_obj"))
        for f in list(d.glob("*.hip")) + list(d.glob("*.h")):
            f.write_text(f.read_text().replace('"../../include/vsg_orb.h"', f'"{ROOT}/include/vsg_orb.h"')
                                        .replace('"../../include/vsg_orb_debug.h"', f'"{ROOT}/include/vsg_orb_debug.h"'))
        core = d / "vsg_octree_core.h"
        s = core.read_text()
        s = sub(s, "namespace vsg {\nnamespace octree {\n", "#ifndef VSG_OCT_STAMP\n#define VSG_OCT_STAMP(tag)\n#endif\n\nnamespace vsg {\nnamespace octree {\n", "namespace")
        s = sub(s, "  int cur = 0;\n  pts.load(g, npts);", "  int cur = 0;\n  VSG_OCT_STAMP(1);\n  pts.load(g, npts);\n  VSG_OCT_STAMP(2);", "load")
        s = sub(s, "  bool finish = false;\n", "  VSG_OCT_STAMP(3);\n  bool finish = false;\n", "initial nodes")
        s = sub(s, "      nL = hist_main_pass(g, W, cur, nL, hist_D, &nV);\n", "      nL = hist_main_pass(g, W, cur, nL, hist_D, &nV);\n      VSG_OCT_STAMP(14);\n", "hist main pass")
        s = sub(s, "  if (hist_D) hist_setup(g, W, cur, nL, pts, npts, hist_D);\n", "  if (hist_D) hist_setup(g, W, cur, nL, pts, npts, hist_D);\n  VSG_OCT_STAMP(13);\n", "hist setup")
        s = sub(s, "      nL = run_main_pass(g, P, W, cur, nL, pts, npts, &nV);\n", "      nL = run_main_pass(g, P, W, cur, nL, pts, npts, &nV);\n      VSG_OCT_STAMP(10);\n", "main pass")
        s = sub(s, "        g.sort_partition_phase(sortbuf, nV, (uint16_t *)W.scanA, (uint16_t *)W.scanB);\n        g.sync();",
                "        VSG_OCT_STAMP(20);\n        g.sort_partition_phase(sortbuf, nV, (uint16_t *)W.scanA, (uint16_t *)W.scanB);\n        g.sync();\n        VSG_OCT_STAMP(21);", "sort")
        s = sub(s, "        int nV2 = 0;\n", "        VSG_OCT_STAMP(22);\n        int nV2 = 0;\n", "careful in")
        s = sub(s, "        nV = nV2;\n", "        VSG_OCT_STAMP(23);\n        nV = nV2;\n", "careful out")
        # inside fused_main_passes: after the counting sweep, the coarser counts, every node-only pass, the table, the relabel
        s = sub(s, "  for (int d = D - 1; d >= 1; d--) {  // level d = sums of four level d + 1 cells", "  VSG_OCT_STAMP(12);\n  for (int d = D - 1; d >= 1; d--) {  // level d = sums of four level d + 1 cells", "f12")
        # inside the one-node-per-thread pass: node read, child counts read, block scan, children written, barrier
        s = sub(s, "  g.sync();\n  pts.for_each(g, npts, [&](uint32_t, int &n) { n = W.cellpos[n]; });", "  g.sync();\n  VSG_OCT_STAMP(15);\n  pts.for_each(g, npts, [&](uint32_t, int &n) { n = W.cellpos[n]; });", "f15")
        s = sub(s, "  g.sync();\n  return nL;\n}\n\n// Points in memory", "  g.sync();\n  VSG_OCT_STAMP(30);\n  return nL;\n}\n\n// Points in memory", "final")
        core.write_text(s)
        k = d / "vsg_kernels.hip"
        s = k.read_text()
        stamp = r'''__device__ unsigned long long g_oct_stamps[8 * 32 * 2];
__device__ int g_oct_idx[8];
__device__ __forceinline__ void vsg_oct_stamp(int tag) {
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8) {
    const int lv = blockIdx.x;
    const int i = g_oct_idx[lv];
    if (i < 32) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
      g_oct_stamps[(lv * 32 + i) * 2] = t;
      g_oct_stamps[(lv * 32 + i) * 2 + 1] = (unsigned long long)tag;
      g_oct_idx[lv] = i + 1;
    }
  }
}
#define VSG_OCT_STAMP(tag) vsg_oct_stamp(tag)
'''
        s = sub(s, '#include "vsg_octree_core.h"', stamp + '#include "vsg_octree_core.h"', "include")
        s = sub(s, "  octree::Work W;\n  octree::carve(W, oct_lds, cap, a.hist_big != 0);\n  BlockGroup g;", "  VSG_OCT_STAMP(0);\n  octree::Work W;\n  octree::carve(W, oct_lds, cap, a.hist_big != 0);\n  BlockGroup g;", "enter")
        if "--sort-rounds" in sys.argv:  # one stamp per round of the introsort replay (each stamp costs the wave ~1 k cycles)
            s = sub(s, "          }\n        }\n        __syncthreads();\n      }\n      return;\n    }\n#endif", "          }\n        }\n        __syncthreads();\n        VSG_OCT_STAMP(50);\n      }\n      return;\n    }\n#endif", "sort round")
        s = sub(s, "  if (threadIdx.x == 0) sel_count[frame * kMaxLevels + level] = n;\n", "  if (threadIdx.x == 0) sel_count[frame * kMaxLevels + level] = n;\n  VSG_OCT_STAMP(31);\n", "exit")
        export = '''extern "C" int vsg_debug_oct_stamps(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_stamps), sizeof(unsigned long long) * 8 * 32 * 2);
  if (reset) {
    int z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_oct_idx), z, sizeof(z));
    hipDeviceSynchronize();
  }
  return 0;
}
'''
        s = sub(s, "void launch_debug_sort(hipStream_t s, uint64_t *d_items, int n) {", export + "void launch_debug_sort(hipStream_t s, uint64_t *d_items, int n) {", "export")
        k.write_text(s)
        out = ROOT / "tools" / "_bin" / f"libvsg_octstamp{suffix}.so"
        out.parent.mkdir(exist_ok=True)
        srcs = ["vsg_kernels.hip", "vsg_orb.hip", "vsg_match.hip", "vsg_grid.hip", "vsg_bow.hip", "vsg_frame.hip", "vsg_ctx.hip", "vsg_shard.hip"]
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w",
               "-mllvm", "-amdgpu-mfma-vgpr-form"] + flags + ["-o", str(out)] + [str(d / x) for x in srcs] + ["-ldl", "-lpthread"]
        subprocess.check_call(cmd, cwd=str(d))
        print("built", out)


if __name__ == "__main__":
    main()
