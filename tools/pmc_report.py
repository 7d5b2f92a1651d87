#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches)."""
import collections, csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc*/*/*counter_collection.csv"
for f in sorted(glob.glob(pat)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        if k.startswith("zd::"):
            n = len(disp[k])
            print(k, "dispatches", n, {c: round(x / n) for c, x in sorted(v.items())})
