mkdir -p gpurun_out/r1i
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for c in 1 4 8 16; do for t in 4 8 16 32; do
  echo "threads=$t chunks=$c $(ZIPC_HIP_HOST_THREADS=$t ZIPC_HIP_HOST_CHUNKS=$c REPS=5 python tools/bench_host_forms.py 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["round_trip_ok"], "def", d["deflate_ms_all"], "inf", d["inflate_ms_all"])')"
done; done | tee gpurun_out/r1i/host_forms_sweep.txt
