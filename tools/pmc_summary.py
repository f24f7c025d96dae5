#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (averages per launch)."""
import collections
import csv
import glob
import sys


def kernel_key(name):
    """'void vsg::k_fast_cells<128, 52, 44>(unsigned char const*, ...)' -> 'vsg::k_fast_cells'."""
    if name.startswith('void '):
        name = name[5:]
    name = name.replace('(anonymous namespace)::', 'vsg::')  # the matcher / frame / BoW kernels live in unnamed namespaces
    name = name.split('(')[0].strip()
    name = name.split('<')[0]
    return name[-26:]


def load(pat):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (f, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    return agg, cnt


def main():
    dirs = sys.argv[1:]
    merged = collections.defaultdict(dict)
    for d in dirs:
        agg, cnt = load(d + '/*/*counter_collection.csv')
        for k in agg:
            for c, v in agg[k].items():
                merged[k][c] = v / max(cnt[k], 1)
    cols = sorted({c for k in merged for c in merged[k]})
    print("%-26s " % "kernel" + " ".join("%14s" % c[-14:] for c in cols))
    for k, d in merged.items():
        if not k.startswith("vsg") and "best2" not in k and k.strip():
            continue
        print("%-26s " % k + " ".join("%14.3f" % (d.get(c, 0) / 1e6) for c in cols))


if __name__ == "__main__":
    main()
