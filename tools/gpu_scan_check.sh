#!/bin/bash
# After a change to lz_match's second form: exact against the oracle (sampled streams) on text, 3-bit symbols, the benchmark's
# symbols with the form forced, `Best on text -- then the same-box A/B of tools/ab_wall.sh against zipc_amd/lib/libzipc_hip_base.so.
#   gpurun -- 'bash tools/gpu_scan_check.sh [tag]'   -> gpurun_out/<tag>/scan_check.txt, scan_ab.txt
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
(for d in text c4 c2; do CHECK=1 DATA=$d REPS=2 timeout 300 python3 tools/exp_wall.py; done
 ZIPC_HIP_MATCH_FORM=2 CHECK=1 DATA=c2 REPS=2 timeout 300 python3 tools/exp_wall.py
 LEVEL=3 N_STREAMS=2048 CHECK=1 DATA=text REPS=2 timeout 300 python3 tools/exp_wall.py
 LEVEL=1 CHECK=1 DATA=text REPS=2 timeout 300 python3 tools/exp_wall.py
 CHECK=1 DATA=corpus N_STREAMS=4096 REPS=2 timeout 300 python3 tools/exp_wall.py) 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $OUT/scan_check.txt
[ -f zipc_amd/lib/libzipc_hip_base.so ] && REPS=3 bash tools/ab_wall.sh "zipc_amd/lib/libzipc_hip_base.so zipc_amd/lib/libzipc_hip.so" "${SHAPES:-text c4 c2}" 2 > $OUT/scan_ab.txt 2>&1
cat $OUT/scan_check.txt $OUT/scan_ab.txt
