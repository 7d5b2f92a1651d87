#!/bin/bash
# the two PMC passes of tools/measure_round.sh alone (C2's step only, no extra legs) + the kernel trace
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
rm -rf "$OUT/pmcF" "$OUT/pmcW" "$OUT/trace"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmcF" -o p --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > "$OUT/pmcF.log" 2>&1 || echo "FETCH_SIZE pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmcW" -o p --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > "$OUT/pmcW.log" 2>&1 || echo "WRITE_SIZE pass failed"
python3 "$ROOT/tools/pmc_report.py" "$OUT/pmc[FW]/*counter_collection.csv" --hbm-json "$ROOT/profiles/${TAG}_hbm_traffic.json" && cp "$ROOT/profiles/${TAG}_hbm_traffic.json" "$OUT/hbm_traffic.json"
python3 "$ROOT/tools/pmc_report.py" "$OUT/pmc[FW]/*counter_collection.csv" > "$OUT/pmc_hbm.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || echo "kernel-trace pass failed"
python3 "$B" > "$OUT/bench.json" 2> "$OUT/bench.err"
head -12 "$OUT/trace/"*kernel_stats.csv | cut -d, -f1-6
