export REPS=3
for len in 1048576 16777216; do
  echo "== LEN $len"
  ZIPC_HIP_INFLATE_BLOCKS_TRACE=1 LEN=$len timeout 200 python tools/exp_inflate_blocks.py 2>&1 | grep -v amdgpu.ids | grep -v "^inflate_by_blocks: src\|token_bad\|chain ok" | cut -c1-650 | uniq
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_limits.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-400
