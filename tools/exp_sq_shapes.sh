#!/bin/bash
# SQ counters of lz_match on the shapes nobody had them for (round 4's review): text at `Default and `Best, 3-bit symbols.
#   gpurun -- 'bash tools/exp_sq_shapes.sh r05'  -> gpurun_out/<tag>/sq_shapes.txt (one line per shape, tools/exp_sq_kernel.sh's)
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
{
  echo "== c2 default";   DATA=c2 LEVEL=2 bash "$ROOT/tools/exp_sq_kernel.sh" zipc_amd/lib/libzipc_hip.so
  echo "== text default"; DATA=text LEVEL=2 bash "$ROOT/tools/exp_sq_kernel.sh" zipc_amd/lib/libzipc_hip.so
  echo "== text best (2048 streams)"; DATA=text LEVEL=3 N_STREAMS=2048 bash "$ROOT/tools/exp_sq_kernel.sh" zipc_amd/lib/libzipc_hip.so
  echo "== c4 default (2048 x 1 MiB)"; DATA=c4 LEVEL=2 bash "$ROOT/tools/exp_sq_kernel.sh" zipc_amd/lib/libzipc_hip.so
} > "$OUT/sq_shapes.txt" 2>&1
cat "$OUT/sq_shapes.txt"
