#!/bin/bash
# One GPU call for a deflate change: the deflate-side -m gpu tests, real text and a short bench line.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests -m gpu -x -q -k "not fullsize" 2>&1 | tail -3
timeout 300 python tools/bench_text.py 2>/dev/null | cut -c1-420
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','deflate_gib_s','inflate_gib_s','c4_deflate_gib_s')}); print({k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
