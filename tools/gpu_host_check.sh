#!/bin/bash
# After a change to the many-stream host forms (api.hip many_streams): their tests, the old library beside the new one
# on the same box (zipc_amd/lib/libzipc_hip_old.so, if one was built), every switch of the forms in turn, and where each
# sub-batch of a call was when.   usage (on the GPU box): bash tools/gpu_host_check.sh <tag>
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG/host_check.txt; : > $O
python3 -m pytest tests/test_gpu_host_batch.py tests/test_gpu_zipc.py -x -q 2>&1 | tail -1 | tee -a $O
echo "ZIPC_HIP_HOST_PACK=0: $(ZIPC_HIP_HOST_PACK=0 python3 -m pytest tests/test_gpu_host_batch.py -x -q 2>&1 | tail -1)" | tee -a $O
row() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); m=lambda x: sorted(x)[len(x)//2]; print(d["round_trip_ok"], "def median %.2f min %.2f max %.2f" % (m(d["deflate_ms_all"]), min(d["deflate_ms_all"]), max(d["deflate_ms_all"])), "| inf median %.2f min %.2f max %.2f" % (m(d["inflate_ms_all"]), min(d["inflate_ms_all"]), max(d["inflate_ms_all"])))'; }
R=${REPS:-11}
for n in ${NS:-4096 16384}; do
  for rep in 1 2 3; do
    if [ -f zipc_amd/lib/libzipc_hip_old.so ]; then
      echo "n=$n old $(ZIPC_HIP_LIB=$ROOT/zipc_amd/lib/libzipc_hip_old.so N_STREAMS=$n REPS=$R python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
    fi
    echo "n=$n new $(N_STREAMS=$n REPS=$R python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
    echo "n=$n new, ZIPC_HIP_HOST_PACK=0 $(ZIPC_HIP_HOST_PACK=0 N_STREAMS=$n REPS=$R python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
  done
  for sw in ${SWITCHES:-ZIPC_HIP_HOST_PACK_WGS=4 ZIPC_HIP_HOST_PACK_WGS=8 ZIPC_HIP_HOST_PACK_WGS=64 ZIPC_HIP_HOST_CHUNKS=3 ZIPC_HIP_HOST_CHUNKS=4 ZIPC_HIP_HOST_CHUNKS=5 ZIPC_HIP_HOST_CHUNKS=6 ZIPC_HIP_HOST_THREADS=4 ZIPC_HIP_HOST_THREADS=16 ZIPC_HIP_HOST_H2D_MIB=0 ZIPC_HIP_HOST_H2D_MIB=4 ZIPC_HIP_HOST_H2D_MIB=64}; do
    echo "n=$n $sw $(env $sw N_STREAMS=$n REPS=$R python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
  done
  ZIPC_HIP_HOST_TIMING=1 N_STREAMS=$n REPS=1 python3 tools/bench_host_forms.py 2>&1 >/dev/null | grep -A6 "zipc_hip" | tail -14 | tee -a $O
done
