export REPS=5
for len in 1048576 16777216; do
  echo "== LEN $len"
  LEN=$len timeout 200 python tools/exp_inflate_blocks.py 2>&1 | grep -v amdgpu.ids | cut -c1-650 | grep -v zeros
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "one_stream" 2>&1 | tail -3
