#!/bin/bash
# Instruction-mix counters (SQ) of every kernel of one C2 deflate+inflate pass, current build.
# Two --pmc passes (the groups that fit the SQ's counter slots).  Output:
# gpurun_out/sq/<pass>/p_counter_collection.csv + gpurun_out/sq/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DEFLATE=${DEFLATE:-1} REPS=2 LEVEL=2 N_STREAMS=${N_STREAMS:-16384} BITS=4
i=0
for ctrs in "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES SQ_WAVE_CYCLES" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
            "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/p$i" -o p --output-format csv \
    -- python3 "$ROOT/tools/exp_inflate.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed rc=$? (see p$i.log)"
done
cd "$ROOT"
python3 tools/pmc_report.py "$OUT/*/*counter_collection.csv" | tee "$OUT/summary.txt"
python3 tools/pmc_report.py "$OUT/*/*counter_collection.csv" --sq-json "$OUT/sq_counters.json"
grep -h '"ms"' "$OUT"/p1.log | tail -1 >> "$OUT/summary.txt"
