#!/bin/bash
# Same-box A/B of lz_match's forms (ZIPC_HIP_MATCH_FORM) on the shapes of tools/exp_wall.py, sampled streams compared with the oracle.
# usage: ab_form.sh "<form1> <form2> ..." "<data1> <data2> ..." [ROUNDS] [LEVEL]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${3:-1}
for r in $(seq $ROUNDS); do
  for data in $2; do
    for form in $1; do
      DATA=$data KERNELS=1 CHECK=1 LEVEL=${4:-2} ZIPC_HIP_MATCH_FORM=$form REPS=${REPS:-3} python3 "$ROOT/tools/exp_wall.py" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-8s form %s %s exact %s L%d defl %8.3f infl %7.3f step %8.3f ratio %.4f | %s' % (d['data'], '$form', 'ok ' if d['ok'] else 'BAD', d['exact'], d['level'], d['deflate_ms'], d['inflate_ms'], d['step_ms'], d['ratio'],
  ' '.join('%s %.3f' % (k.replace('deflate_','').replace('crc32_','crc_').replace('inflate_batch','infl'), v) for k, v in sorted(d['kernels_ms'].items()))))"
    done
  done
done
