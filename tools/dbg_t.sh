export REPS=5
TRIALS=60 SEED=71 timeout 600 python tools/fuzz_inflate_blocks.py 2>&1 | tail -1
for len in 1048576 16777216; do
    LEN=$len timeout 400 python tools/exp_inflate_blocks.py 2>&1 | grep -v amdgpu.ids | grep -v "zeros" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms']; print(d['input'][:20], d['encoder'], d['blocks'], d['inflate_ms'], 'dry', k.get('inflate_blocks_dry'), 'tok', k.get('inflate_blocks_token'), 'res', k.get('inflate_resolve'))
    elif 'MISMATCH' in l: print(l.strip()[:200])"
done
