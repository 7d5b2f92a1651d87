#!/bin/bash
# C4's OWN counter passes (the review: "bench.py --config c4 prices C4 with C2's counters"): FETCH_SIZE / WRITE_SIZE and
# the SQ instruction counters of `bench.py --config c4` (8192 members x 1 MiB of 3-bit symbols, `Default), each in a
# run of its own with --kernel-trace only, then the line that quotes them.
#   gpurun -- 'bash tools/measure_c4_counters.sh r06'   -> profiles/r06_hbm_traffic_c4.json, r06_sq_counters_c4.json, gpurun_out/r06/c4/*
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG/c4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
ARGS="--config c4 --no-cpu-baseline --no-archive-check --steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmcF" -o p --output-format csv -- python3 "$B" $ARGS > "$OUT/pmcF.log" 2>&1 || echo "FETCH_SIZE pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmcW" -o p --output-format csv -- python3 "$B" $ARGS > "$OUT/pmcW.log" 2>&1 || echo "WRITE_SIZE pass failed"
python3 "$ROOT/tools/pmc_report.py" "$OUT/pmc[FW]/*counter_collection.csv" --workload c4 --hbm-json "$ROOT/profiles/${TAG}_hbm_traffic_c4.json" && cp "$ROOT/profiles/${TAG}_hbm_traffic_c4.json" "$OUT/"
i=0
for ctrs in "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
            "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/sq$i" -o p --output-format csv -- python3 "$B" $ARGS > "$OUT/sq$i.log" 2>&1 || echo "SQ pass $i failed"
done
python3 "$ROOT/tools/pmc_report.py" "$OUT/sq*/*counter_collection.csv" --workload c4 --sq-json "$ROOT/profiles/${TAG}_sq_counters_c4.json" && cp "$ROOT/profiles/${TAG}_sq_counters_c4.json" "$OUT/"
python3 "$ROOT/tools/pmc_report.py" "$OUT/sq*/*counter_collection.csv" > "$OUT/sq_counters_c4.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$B" $ARGS > "$OUT/trace_bench_c4.json" 2> "$OUT/trace.err" || echo "kernel-trace pass failed"
cd "$ROOT"
python3 "$B" --config c4 --steps 3 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 -c "
import json
d=json.load(open('$OUT/bench_c4.json'))
print({k:d[k] for k in ('value','ms_per_step')}); print({k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))}); print(d.get('archive_check'))"
head -8 "$OUT/trace/"*kernel_stats.csv | cut -d, -f1-6
