#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).

  pmc_report.py "<glob>"                  human-readable lines
  pmc_report.py "<glob>" --hbm-json OUT   per-kernel HBM traffic per launch from the
                                          FETCH_SIZE / WRITE_SIZE passes (both in KiB,
                                          MI355X_MICROARCH.md section HBM)
  pmc_report.py "<glob>" --sq-json OUT    per-kernel instruction counts per launch from the
                                          SQ passes (tools/exp_sq_counters.sh); bench.py turns
                                          them into the instruction-issue bound
"""
import collections, csv, glob, json, sys

_vals = {sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a in ("--hbm-json", "--sq-json", "--workload")}
args = [a for a in sys.argv[1:] if not a.startswith("--") and a not in _vals]
pat = args[0] if args else "gpurun_out/pmc*/*/*counter_collection.csv"
hbm_out = sys.argv[sys.argv.index("--hbm-json") + 1] if "--hbm-json" in sys.argv else None
sq_out = sys.argv[sys.argv.index("--sq-json") + 1] if "--sq-json" in sys.argv else None
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
    name_ = k.replace("zd::", "").replace("_kernel", "").replace("_window", "").replace("_streams", "")
    if sq_out is not None and "SQ_INSTS_VALU" in mean:
        out[name_] = {c: mean[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
                                           "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES") if c in mean}
        continue
    if sq_out is not None:
        continue
    if hbm_out is None:
        print(k, "dispatches", {c: len(disp[k][c]) for c in mean}, {c: round(x) for c, x in mean.items()})
    else:
        name = k.replace("zd::", "").replace("_kernel", "").replace("_window", "")
        fetch = mean.get("FETCH_SIZE", 0.0) * 1024
        write = mean.get("WRITE_SIZE", 0.0) * 1024
        out[name] = {"fetch_bytes": fetch, "write_bytes": write, "bytes": fetch + write}
workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "c2"
if sq_out:
    what = ("tools/exp_inflate.py on C2 (tools/exp_sq_counters.sh: 16 384 streams in two slices, a launch covers half the batch)" if workload == "c2" else
            "bench.py --config c4 (tools/measure_c4_counters.sh: 8192 members of 1 MiB in two slices, a launch covers half the batch)")
    json.dump({"note": "mean wave-instructions per launch, rocprofv3 --pmc SQ passes of " + what + "; issue bounds: "
                       "vector = VALU x 4 clocks / (1024 SIMDs x f), scalar = (SALU + BRANCH + SMEM) / (256 CUs x f), f = shader clock",
               "workload": workload,
               "launch_share": 0.5,
               "kernels": out}, open(sq_out, "w"), indent=1)
if hbm_out:
    json.dump({"note": "bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes; on gfx950 "
                       "FETCH_SIZE reads half the bytes of a 16 B/lane coalesced stream and is uncalibrated for "
                       "narrower accesses (MI355X_MICROARCH.md): raw counter values are kept here",
               "workload": workload,
               "kernels": out}, open(hbm_out, "w"), indent=1)
