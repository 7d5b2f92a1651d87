#!/bin/bash
# the by-blocks inflate of calls of long streams: fuzz against the oracle (single streams, then batches), then timings
mkdir -p gpurun_out
{
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
TRIALS=${TRIALS:-60} SEED=11 timeout 600 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -5
TRIALS=${CALLS:-30} SEED=12 BATCH=12 timeout 900 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -8
ZIPC_HIP_INFLATE_FOLLOW=1 TRIALS=${CALLS:-30} SEED=13 BATCH=12 timeout 900 python3 tools/fuzz_inflate_blocks.py 2>&1 | tail -8
timeout 300 python3 tools/exp_inflate_many.py 2>&1 | tail -12
ZIPC_HIP_INFLATE_BLOCKS=0 REPS=3 timeout 300 python3 tools/exp_inflate_many.py 2>&1 | tail -12
N=8 LEN=8388608 REPS=3 timeout 300 python3 tools/exp_inflate_many.py 2>&1 | tail -12
N=1 LEN=67108864 REPS=3 timeout 300 python3 tools/exp_inflate_many.py 2>&1 | tail -12
} > gpurun_out/inflate_many.log 2>&1
grep -v amdgpu.ids gpurun_out/inflate_many.log | tail -60
