#!/usr/bin/env python3
"""Static VALU instruction mix of the extractor / matcher kernels (gfx950 ISA of the sources as they are, compiled with
the library's flags): how many of a kernel's vector instructions are on the full-rate path (plain 32-bit add / sub / logic
/ shift-right / move and fp32 add / mul / fma: 2.3 cycles per wave64 instruction per SIMD, profiles/r01_c_ubench_valu_rates.txt)
and how many on the half-rate one (VOP3 integer, packed 16-bit, dot, perm, alignbyte, multiplies, min / max, converts,
SDWA / DPP forms: 4.2 cycles).  STATIC counts: loops are not weighted by their trip counts, so the priced cycles per
instruction are an estimate between the two rates, not a measurement.   python tools/isa_mix.py > profiles/rNN_isa_mix.json"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from source_hash import source_hash  # noqa: E402

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_min_i16", "v_add_u16", "v_sub_u16", "v_not_b32",
        "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mac_f32", "v_add_co_u32", "v_sub_co_u32", "v_cndmask_b32",
        "v_bitop3_b32"}  # v_bitop3_b32: 2.3 cycles measured (tools/ubench.hip)
KERNELS = {"k_pyramid": "pyramid", "k_fast_cells": "fast", "k_blur": "blur", "k_octree": "octree", "k_orient_desc": "orient_desc",
           "k_block_best2_mfma": "match"}


def main():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-mllvm",
             "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S"]
    out = {"source_hash": source_hash(), "rates_cycles": {"full": 2.3, "half": 4.2},
           "_comment": __doc__.split("python tools")[0].strip()}
    with tempfile.TemporaryDirectory() as td:
        for src in ("vsg_kernels.hip", "vsg_match.hip"):
            asm = Path(td) / (src + ".s")
            subprocess.run(["hipcc"] + flags + ["-o", str(asm), str(ROOT / "visual_sgraphs_amd" / "csrc" / src)],
                           check=True, stderr=subprocess.DEVNULL, cwd=str(ROOT / "visual_sgraphs_amd" / "csrc"))
            name, counts = None, None
            for line in asm.read_text().splitlines():
                m = re.match(r"^(_Z\w+):\s", line + " ")
                if m:
                    name = next((v for k, v in KERNELS.items() if k in m.group(1) and "_v2" not in m.group(1)
                                 and "octree_blur" not in m.group(1)), None)
                    if name == "orient_desc" and "ILb1E" in m.group(1):
                        name = None  # the mirroring variant of the latency path
                    if name == "octree" and "k_octree_fewILi1E" not in m.group(1):
                        name = None  # the instantiation the serialised C2 step launches (k_octree_few<1>)
                    if name == "fast" and "Li128ELi52ELi52E" not in m.group(1):
                        name = None  # the tile-pitch class of 640x480
                    counts = {"full": 0, "half": 0, "dpp_sdwa": 0} if name else None
                    continue
                if counts is None:
                    continue
                t = line.strip().split()
                if not t:
                    continue
                if t[0] == "s_endpgm":
                    total = counts["full"] + counts["half"]
                    out[name] = dict(counts, valu_static=total,
                                     priced_cycles_per_inst=round((2.3 * counts["full"] + 4.2 * counts["half"]) / max(total, 1), 3))
                    counts = None
                    continue
                if not t[0].startswith("v_") or t[0].startswith("v_mfma") or t[0].startswith("v_readlane") or t[0].startswith("v_readfirstlane"):
                    continue
                op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", t[0])
                slow_form = t[0].endswith("_dpp") or t[0].endswith("_sdwa") or t[0].endswith("_e64")
                if slow_form and (t[0].endswith("_dpp") or t[0].endswith("_sdwa")):
                    counts["dpp_sdwa"] += 1
                counts["full" if (op in FULL and not t[0].endswith("_sdwa") and not t[0].endswith("_e64")) else "half"] += 1
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
