#!/bin/bash
# Same-box A/B of builds of libzipc_hip.so under one ZIPC_HIP_MATCH_FORM: usage ab_libs_form.sh FORM "<lib1.so> ..." "<data> ..." [ROUNDS]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in $(seq ${4:-1}); do for data in $3; do for lib in $2; do
  DATA=$data KERNELS=1 ZIPC_HIP_MATCH_FORM=$1 ZIPC_HIP_LIB="$ROOT/$lib" REPS=${REPS:-3} python3 "$ROOT/tools/exp_wall.py" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-8s %-28s %s defl %8.3f | lz_match %.3f lz_parse %.3f' % (d['data'], d['lib'], 'ok ' if d['ok'] else 'BAD', d['deflate_ms'], d['kernels_ms'].get('lz_match',0), d['kernels_ms'].get('lz_parse',0)))"
done; done; done
