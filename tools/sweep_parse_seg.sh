#!/bin/bash
# lz_parse by segments: the segment size against the stream length (tools/bench_single.py, one stream alone)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for len in 1048576 16777216; do
  for seg in 4096 8192 16384 32768 65536; do
    LEN=$len ZIPC_HIP_PARSE_SEG=$seg python3 "$ROOT/tools/bench_single.py" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); k=d['level 2']['kernels_ms']
    print('len $len seg $seg %-28s defl %7.3f | spec %.3f stitch %.3f gather %.3f | plan %.3f scan %.3f pack %.3f | chain %.3f match %.3f' % (d['input'][-16:], d['level 2']['gpu_deflate_ms'], k.get('lz_parse_spec',0), k.get('lz_parse_stitch',0), k.get('lz_parse_gather',0), k.get('deflate_plan',0), k.get('deflate_scan',0), k.get('deflate_pack',0), k['lz_chain'], k['lz_match']))"
  done
done
