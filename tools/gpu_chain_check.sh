#!/bin/bash
# One GPU call for a change to lz_chain: randomized parity (many short streams, long streams by segments), the
# shapes of tools/exp_wall.py with the exchange kernel and with the peel kernel (ZIPC_HIP_CHAIN=peel).
cd "${GRAFT_REPO_ROOT:-.}"
SEED=${SEED:-93000}
echo "== fuzz"; timeout 600 python3 tools/fuzz_gpu.py $SEED ${NSEEDS:-2} 2>&1 | tail -2
echo "== fuzz, one wave per stream forms"; ZIPC_HIP_PARSE_SEGMENTS=0 timeout 600 python3 tools/fuzz_gpu.py $SEED 1 2>&1 | tail -1
echo "== long streams"; timeout 600 python3 tools/fuzz_long.py $SEED 1 2>&1 | tail -1
for data in ${DATAS:-c2 text c4}; do
  for chain in xchg peel; do
    echo "== $data chain=$chain"
    ZIPC_HIP_CHAIN=$chain DATA=$data CHECK=1 KERNELS=1 REPS=3 timeout 600 python3 tools/exp_wall.py 2>&1 | tail -1 | cut -c1-900
  done
done
echo "== one stream"; timeout 300 python3 tools/bench_single.py 2>/dev/null | cut -c1-400
