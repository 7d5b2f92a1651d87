#!/bin/bash
# Where does inflate_batch's L2->fabric traffic come from?  FETCH_SIZE / WRITE_SIZE and the
# raw L2 counters of the inflate kernel alone (tools/exp_inflate.py, DEFLATE=0 in the timed
# loop) over stream counts whose per-XCD output does / does not fit the 4 MiB L2, and on
# incompressible input (stored blocks: no match copies).  One --pmc pass per counter group.
# Output: gpurun_out/traffic/<config>/<pass>/p_counter_collection.csv + summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DEFLATE=0 REPS=3 LEVEL=2
run_cfg() {  # name n_streams bits
  export N_STREAMS=$2 BITS=$3
  for pass in "F:FETCH_SIZE" "W:WRITE_SIZE" "H:TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
              "R:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
    tag=${pass%%:*}; ctrs=${pass#*:}
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/$1/$tag" -o p --output-format csv \
      -- python3 "$ROOT/tools/exp_inflate.py" > "$OUT/$1/$tag.log" 2>&1 || echo "pass $1/$tag failed rc=$?"
  done
}
for n in 256 1024 4096 16384; do mkdir -p "$OUT/c2_n$n"; run_cfg c2_n$n $n 4; done
mkdir -p "$OUT/c5_n16384"; run_cfg c5_n16384 16384 8
cd "$ROOT"
for d in "$OUT"/*/; do
  echo "== $(basename "$d") $(grep -h -o '"ratio": [0-9.]*' "$d/F.log" | head -1)"
  python3 tools/pmc_report.py "$d/*/*counter_collection.csv" | grep inflate_batch
done | tee "$OUT/summary.txt"
