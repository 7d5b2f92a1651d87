#!/bin/bash
# Same-box A/B of builds of libzipc_hip.so: wall and per-kernel ms on the shapes of tools/exp_wall.py.
# usage: ab_wall.sh "<lib1.so> <lib2.so> ..." "<data1> <data2> ..." [ROUNDS]   (paths relative to the repo)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for data in $2; do
    for lib in $1; do
      DATA=$data KERNELS=1 ZIPC_HIP_LIB="$ROOT/$lib" REPS=${REPS:-3} python3 "$ROOT/tools/exp_wall.py" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-8s %-26s %s L%d defl %8.3f infl %7.3f step %8.3f ratio %.4f | %s' % (d['data'], d['lib'], 'ok ' if d['ok'] else 'BAD', d['level'], d['deflate_ms'], d['inflate_ms'], d['step_ms'], d['ratio'],
  ' '.join('%s %.3f' % (k.replace('deflate_','').replace('crc32_','crc_').replace('inflate_batch','infl'), v) for k, v in sorted(d['kernels_ms'].items()))))"
    done
  done
done
