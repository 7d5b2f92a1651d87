#!/bin/bash
# A long randomized parity run on the build at hand, every path against the oracle: batches of assorted streams both
# ways (fuzz_gpu.py), long streams through the many-wave deflate forms (fuzz_long.py), one stream and calls of streams
# through the block path of inflate (fuzz_inflate_blocks.py).  SEED0 shifts every seed.
mkdir -p gpurun_out
S=${SEED0:-4000}
{
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 900 python3 tools/fuzz_gpu.py $S 12 2>&1 | tail -3
timeout 900 python3 tools/fuzz_long.py $((S+1)) 6 2>&1 | tail -2
for k in 1 2 3; do TRIALS=150 SEED=$((S+10+k)) timeout 900 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -2; done
for k in 1 2 3; do TRIALS=40 BATCH=16 SEED=$((S+20+k)) timeout 900 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -2; done
ZIPC_HIP_INFLATE_FOLLOW=1 TRIALS=40 BATCH=16 SEED=$((S+30)) timeout 900 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz_campaign.log
cat gpurun_out/fuzz_campaign.log
