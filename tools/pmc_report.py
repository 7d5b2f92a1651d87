#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).

  pmc_report.py "<glob>"                  human-readable lines
  pmc_report.py "<glob>" --hbm-json OUT   per-kernel HBM traffic per launch from the
                                          FETCH_SIZE / WRITE_SIZE passes (both in KiB,
                                          MI355X_MICROARCH.md section HBM)
"""
import collections, csv, glob, json, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = args[0] if args else "gpurun_out/pmc*/*/*counter_collection.csv"
hbm_out = sys.argv[sys.argv.index("--hbm-json") + 1] if "--hbm-json" in sys.argv else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(pat)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
out = {}
for k, v in agg.items():
    if not k.startswith("zd::"):
        continue
    mean = {c: x / len(disp[k][c]) for c, x in sorted(v.items())}
    if hbm_out is None:
        print(k, "dispatches", {c: len(disp[k][c]) for c in mean}, {c: round(x) for c, x in mean.items()})
    else:
        name = k.replace("zd::", "").replace("_kernel", "").replace("_window", "")
        fetch = mean.get("FETCH_SIZE", 0.0) * 1024
        write = mean.get("WRITE_SIZE", 0.0) * 1024
        out[name] = {"fetch_bytes": fetch, "write_bytes": write, "bytes": fetch + write}
if hbm_out:
    json.dump({"note": "bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes; on gfx950 "
                       "FETCH_SIZE reads half the bytes of a 16 B/lane coalesced stream and is uncalibrated for "
                       "narrower accesses (MI355X_MICROARCH.md): raw counter values are kept here",
               "kernels": out}, open(hbm_out, "w"), indent=1)
