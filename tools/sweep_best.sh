#!/bin/bash
# The second form's schedule constants at level `Best (and `Default) on text and the corpus, one variant build each
# (tools/build_variant.sh NAME "-DZD_SCAN_..."): does a walk of hundreds of candidates want other values than `Default's 33?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LIBS="zipc_amd/lib/libzipc_hip.so"
for v in dh16 dh64 dh128 mw8 mw32 ho16 ho32 ho48 cm8 cm16 cm24 cm40 np3 np4; do [ -f zipc_amd/lib/libzipc_hip_$v.so ] && LIBS="$LIBS zipc_amd/lib/libzipc_hip_$v.so"; done
for lvl in 3 2; do
  for data in text corpus; do
    for lib in $LIBS; do
      N_STREAMS=2048 LEVEL=$lvl DATA=$data KERNELS=1 ZIPC_HIP_LIB="$ROOT/$lib" REPS=2 python3 tools/exp_wall.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-7s L%d %-24s %s defl %9.3f ms  lz_match %9.3f' % (d['data'], d['level'], d['lib'], 'ok ' if d['ok'] else 'BAD', d['deflate_ms'], d['kernels_ms'].get('lz_match', 0)))"
    done
  done
done
