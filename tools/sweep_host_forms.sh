#!/bin/bash
# PCIe-inclusive rate of the many-stream host forms over the number of sub-batches (and host
# threads), one box: tools/bench_host_forms.py, REPS timed calls per direction each (median, min, max).
mkdir -p gpurun_out/hostsweep
for c in ${CHUNKS:-1 2 4 8}; do for t in ${THREADS:-8}; do
  echo "threads=$t sub-batches=$c $(ZIPC_HIP_HOST_THREADS=$t ZIPC_HIP_HOST_CHUNKS=$c REPS=${REPS:-5} python tools/bench_host_forms.py 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); m=lambda x: sorted(x)[len(x)//2]; print(d["round_trip_ok"], "def median %.1f min %.1f max %.1f" % (m(d["deflate_ms_all"]), min(d["deflate_ms_all"]), max(d["deflate_ms_all"])), "| inf median %.1f min %.1f max %.1f" % (m(d["inflate_ms_all"]), min(d["inflate_ms_all"]), max(d["inflate_ms_all"])))')"
done; done | tee gpurun_out/hostsweep/host_forms_sweep.txt
