#!/bin/bash
# The round's measurement set, one command: bench.py (default flags, with cpu_baseline), the
# rocprofv3 --kernel-trace --stats summary of the same command, the FETCH_SIZE / WRITE_SIZE
# PMC passes (separate runs), bench.py under torch.distributed.run with one process, the
# other BASELINE configs and the PCIe-inclusive host forms.  Run on the GPU box:
#   gpurun -- 'bash tools/measure_round.sh r02'      -> gpurun_out/r02/*  (TAG names the profiles/ files too)
# The PMC passes write profiles/${TAG}_hbm_traffic.json on the box BEFORE the final bench.py
# run, so that run's roofline.traffic comes from counters of the same build; the file is
# also copied to the output directory.
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmcF" -o p --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > "$OUT/pmcF.log" 2>&1 || echo "FETCH_SIZE pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmcW" -o p --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > "$OUT/pmcW.log" 2>&1 || echo "WRITE_SIZE pass failed"
python3 "$ROOT/tools/pmc_report.py" "$OUT/pmc[FW]/*counter_collection.csv" --hbm-json "$ROOT/profiles/${TAG}_hbm_traffic.json" && cp "$ROOT/profiles/${TAG}_hbm_traffic.json" "$OUT/hbm_traffic.json"
python3 "$ROOT/tools/pmc_report.py" "$OUT/pmc[FW]/*counter_collection.csv" > "$OUT/pmc_hbm.txt"
# the SQ counter pass too comes BEFORE the bench line it is quoted in (roofline.issue_bound reads profiles/*sq_counters.json)
(cd "$ROOT" && DEFLATE=1 bash tools/exp_sq_counters.sh > "$OUT/sq.log" 2>&1; cp gpurun_out/sq/sq_counters.json "$OUT/sq_counters.json"; cp gpurun_out/sq/summary.txt "$OUT/sq_counters.txt"; cp gpurun_out/sq/sq_counters.json "profiles/${TAG}_sq_counters.json")
python3 "$B" > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || echo "kernel-trace pass failed"
python3 "$B" --alone-pass --no-cpu-baseline --no-extra-legs > "$OUT/bench_alone.json" 2> /dev/null || echo "alone pass failed"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 "$B" --gpus 1 --no-cpu-baseline > "$OUT/torchrun.json" 2> "$OUT/torchrun.err" || echo "torchrun failed"
cd "$ROOT"
python3 tools/bench_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
python3 tools/bench_host_forms.py > "$OUT/host_forms.json" 2> /dev/null
python3 tools/bench_text.py > "$OUT/real_text.json" 2> /dev/null
python3 tools/bench_single.py > "$OUT/single_stream.jsonl" 2> /dev/null
LEN=16777216 python3 tools/bench_single.py >> "$OUT/single_stream.jsonl" 2> /dev/null
ZIPC_HIP_PARSE_SEGMENTS=0 ZIPC_HIP_INFLATE_BLOCKS=0 python3 tools/bench_single.py > "$OUT/single_stream_one_wave.jsonl" 2> /dev/null
# inflate of one stream by a wave per block: streams of the reference's encoder and of zlib, 1 MiB and 16 MiB, and by its one wave
python3 tools/exp_inflate_blocks.py > "$OUT/inflate_blocks.jsonl" 2> /dev/null
LEN=16777216 python3 tools/exp_inflate_blocks.py >> "$OUT/inflate_blocks.jsonl" 2> /dev/null
LEN=67108864 REPS=3 python3 tools/exp_inflate_blocks.py >> "$OUT/inflate_blocks.jsonl" 2> /dev/null
ZIPC_HIP_INFLATE_BLOCKS=0 python3 tools/exp_inflate_blocks.py > "$OUT/inflate_blocks_one_wave.jsonl" 2> /dev/null
ZIPC_HIP_INFLATE_BLOCKS=0 LEN=16777216 python3 tools/exp_inflate_blocks.py >> "$OUT/inflate_blocks_one_wave.jsonl" 2> /dev/null
# a call of N long streams: by blocks side by side, by their one waves, and kernel by kernel
{ REPS=5 python3 tools/exp_inflate_many.py; N=8 LEN=8388608 REPS=3 python3 tools/exp_inflate_many.py; N=16 REPS=5 python3 tools/exp_inflate_many.py;
  echo "-- ZIPC_HIP_INFLATE_BLOCKS=0"; ZIPC_HIP_INFLATE_BLOCKS=0 REPS=3 python3 tools/exp_inflate_many.py; } 2> /dev/null | grep -v amdgpu.ids > "$OUT/inflate_many.txt"
CFGS="64x1048576 8x8388608" bash tools/prof_inflate_many.sh 2> /dev/null | grep -v "^[WE]2026" > "$OUT/inflate_many_kernels.txt"
python3 "$B" --config c4 --steps 3 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
echo "== bench"; python3 -c "
import json,sys
d=json.load(open('$OUT/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','deflate_gib_s','inflate_gib_s')}); print(d['roofline']); print(d['cpu_baseline']); print({k:round(v,3) for k,v in d['kernels_ms_per_step'].items()}); print({k:d.get(k) for k in ('e2e_gib_s','text_gib_s','c4_deflate_gib_s','c4_inflate_gib_s')})"
echo "== rocprofv3 kernel stats (same command)"; head -12 "$OUT/trace/"*kernel_stats.csv | cut -d, -f1-6
echo "== torchrun"; cut -c1-200 "$OUT/torchrun.json"
echo "== configs"; cut -c1-260 "$OUT/configs.jsonl"
echo "== host forms"; cat "$OUT/host_forms.json"
