#!/usr/bin/env python3
"""rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; one directory each) -> profiles/traffic_rNN.json in
the form bench.py attaches to its `roofline` object: bytes / instructions PER LAUNCH of every stage kernel.
    python tools/make_traffic_json.py <tag e.g. C2/512> <out.json> <pmc_dir> [<pmc_dir> ...]
Units as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are reported in KB -> x 1024; on
gfx950 FETCH_SIZE counts 64 B per 128-B request of wide (16 B / lane) streaming reads, i.e. half the bytes.  The RAW
values are stored here; bench.py doubles FETCH_SIZE for the kernels whose global reads are 16 B per lane (FAST,
pyramid) and keeps the raw value -- a lower bound -- for the 4-8 B per lane readers (blur, orient+desc, octree)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_summary import load  # noqa: E402
from source_hash import source_hash  # noqa: E402

STAGE = {"k_pyramid": "pyramid", "k_fast_cells": "fast", "k_blur": "blur", "k_octree": "octree", "k_slots": "slots",
         "k_orient_desc": "orient_desc", "best2": "match"}


def main():
    tag, out = sys.argv[1], Path(sys.argv[2])
    merged = {}
    for d in sys.argv[3:]:
        agg, cnt = load(d + "/*/*counter_collection.csv")
        for k in agg:
            stage = next((v for s, v in STAGE.items() if s in k), None)
            if k.strip() == "" or stage is None:
                stage = "match" if k.strip() == "" else stage  # (older dumps: the MFMA matcher's long name was cut to '')
            if stage is None:
                continue
            for c, v in agg[k].items():
                merged.setdefault(stage, {})[c] = v / max(cnt[k], 1)
    res = {}
    for stage, d in merged.items():
        e = {}
        if "FETCH_SIZE" in d:
            e["fetch_bytes"] = int(d["FETCH_SIZE"] * 1024)
        if "WRITE_SIZE" in d:
            e["write_bytes"] = int(d["WRITE_SIZE"] * 1024)
        if "SQ_INSTS_VALU" in d:
            e["valu_insts"] = int(d["SQ_INSTS_VALU"])
        # SQ occupancy / wait counters (quad-cycles summed over waves; MI355X_MICROARCH.md "SQ": WAIT_ANY + WAIT_INST_ANY +
        # ACTIVE_INST_ANY ~ WAVE_CYCLES) and instruction counts, when their passes were run
        for c, key in (("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_WAVES", "waves"),
                       ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_BUSY_CYCLES", "busy_cycles"), ("SQ_WAIT_ANY", "wait_any"),
                       ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                       ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds"),
                       ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_accesses"),
                       ("TCP_TCC_READ_REQ_sum", "l1_to_l2_read_req"), ("TCP_PENDING_STALL_CYCLES_sum", "l1_pending_stall_cycles")):
            if c in d:
                e[key] = int(d[c])
        res[stage] = e
    doc = json.loads(out.read_text()) if out.exists() else {
        "_comment": "HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), raw "
                    "counter values in bytes (counter unit KB x 1024); gfx950 caveat (MI355X_MICROARCH.md): FETCH_SIZE "
                    "under-reports wide (16 B/lane) streaming reads by 2x and is uncalibrated for the 4-12 B per lane "
                    "loads used here, so fetch_bytes is a lower bound; Infinity-Cache hits are counted. valu_insts = "
                    "SQ_INSTS_VALU per launch.  Collected under VSG_NO_OVERLAP=1 (tools/profile_round.sh): octree and blur "
                    "are the SEPARATE-LAUNCH forms k_octree / k_blur, not the fused k_octree_blur of the timed region."}
    doc[tag] = res
    doc["source_hash"] = source_hash()  # tools/source_hash.py: the sources these counters were measured on
    out.write_text(json.dumps(doc, indent=1))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
