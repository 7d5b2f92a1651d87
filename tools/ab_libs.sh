#!/bin/bash
# Same-box A/B of two builds of libzipc_hip.so on C2 (tools/exp_inflate.py, HIP-event
# per-kernel ms): alternates base/new ROUNDS times so drift shows up as spread.
# usage: ab_libs.sh <base.so> <new.so> [ROUNDS]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${3:-3}
for r in $(seq $ROUNDS); do
  for which in "$1" "$2"; do
    ZIPC_HIP_LIB="$ROOT/$which" REPS=5 python3 "$ROOT/tools/exp_inflate.py" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['ms']
print('$which'.split('/')[-1], 'ok' if d['roundtrip_ok'] else 'ROUNDTRIP FAILED', 'ratio %.6f' % d['ratio'], ' '.join('%s %.3f' % (k.replace('_kernel','').replace('zd::',''), v) for k, v in sorted(m.items())))"
  done
done
