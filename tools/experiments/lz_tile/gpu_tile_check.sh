#!/bin/bash
# One GPU call for a change to lz_tile.hip: randomized parity with the tile kernel forced onto every stream above
# 8 KiB (ZIPC_HIP_PARSE_SEGMENTS=0 keeps the one-wave-per-stream forms it lives among), the shapes of
# tools/exp_wall.py against the old kernels (ZIPC_HIP_TILE=0) with sampled streams compared with the oracle.
cd "${GRAFT_REPO_ROOT:-.}"
SEED=${SEED:-91000}
echo "== fuzz (tile kernel on every stream above 8 KiB)"
ZIPC_HIP_PARSE_SEGMENTS=0 timeout 600 python3 tools/fuzz_gpu.py $SEED ${NSEEDS:-3} 2>&1 | tail -4
echo "== fuzz, every stream left to the old kernels"
ZIPC_HIP_PARSE_SEGMENTS=0 ZIPC_HIP_TILE_PUNT=1 timeout 600 python3 tools/fuzz_gpu.py $SEED 1 2>&1 | tail -2
echo "== long streams"
ZIPC_HIP_PARSE_SEGMENTS=0 timeout 600 python3 tools/fuzz_long.py $SEED 1 2>&1 | tail -3
for data in ${DATAS:-c2 text c4}; do
  for tile in 1 0; do
    echo "== $data tile=$tile"
    ZIPC_HIP_TILE=$tile DATA=$data CHECK=1 KERNELS=1 REPS=3 timeout 600 python3 tools/exp_wall.py 2>&1 | tail -1 | cut -c1-900
  done
done
