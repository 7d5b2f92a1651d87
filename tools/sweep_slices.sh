#!/bin/bash
# Wall time of the batch forms against the number of slices (queues) a batch call uses.
#   gpurun -- 'bash tools/sweep_slices.sh'   -> gpurun_out/slices.jsonl
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/slices.jsonl
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
for data in c2 text c4; do
  for k in ${SLICES:-1 2 3 4 6 8}; do
    DATA=$data ZIPC_HIP_SLICES=$k REPS=${REPS:-5} python3 "$ROOT/tools/exp_wall.py" 2>/dev/null | tee -a "$OUT" | cut -c1-330
  done
done
