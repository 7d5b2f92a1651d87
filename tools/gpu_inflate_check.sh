#!/bin/bash
# One GPU call for an inflate change: the -m gpu suite, the phase clocks of the timing build
# (built beforehand with tools/build_timing_lib.sh ZD_INFLATE_PHASES) and a short bench line.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ZIPC_HIP_PHASE_LIB=zipc_amd/csrc/build/timing_ZD_INFLATE_PHASES.so timeout 300 python tools/exp_inflate_phases.py 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','deflate_gib_s','inflate_gib_s')}); print({k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
