export REPS=3
for len in 16777216; do
  echo "== LEN $len"
  ZIPC_HIP_INFLATE_BLOCKS_TRACE=1 LEN=$len timeout 200 python tools/exp_inflate_blocks.py 2>&1 | grep -v amdgpu.ids | grep -v "^inflate_by_blocks: src\|token_bad" | cut -c1-650 | uniq
done
