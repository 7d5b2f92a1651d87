#!/bin/bash
# SQ counters of ONE kernel on one of tools/exp_wall.py's shapes, for one or more builds:
#   KERNEL=lz_match DATA=c4 tools/exp_sq_kernel.sh lib1.so [lib2.so ...]   (paths relative to the repo)
# Two --pmc passes per build (issue mix, LDS), rocprofv3 with --kernel-trace only (no other trace domains).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KERNEL=${KERNEL:-lz_match}
OUT=$ROOT/gpurun_out/sqk
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DATA=${DATA:-c2} REPS=1 ZIPC_HIP_SLICES=1
for lib in "$@"; do
  tag=$(basename "$lib" .so)_$DATA
  export ZIPC_HIP_LIB="$ROOT/$lib"
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rm -rf "$OUT/$tag.p$i"
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/$tag.p$i" -o p --output-format csv \
      -- python3 "$ROOT/tools/exp_wall.py" > "$OUT/$tag.p$i.log" 2>&1 || echo "pass $i of $tag failed"
  done
  python3 - "$OUT" "$tag" "$KERNEL" <<'PY'
import collections, csv, glob, sys
out, tag, kern = sys.argv[1:4]
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob("%s/%s.p*/**/*counter_collection.csv" % (out, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
m = {c: agg[c] / len(n[c]) for c in agg}
g = lambda c: m.get(c, 0.0)
print("%-28s %s: VALU %.2fG SALU %.2fG BR %.2fG LDS %.2fG VMEM_WR %.1fM | wave_cycles %.1fG wait_any %.0f%% wait_inst %.0f%% active %.0f%% | "
      "LDS busy/CU %.2fM bank_conf %.2fM addr_conf %.2fM  valu_ms@2.38GHz %.1f scalar_ms %.1f" % (
      tag, kern, g("SQ_INSTS_VALU") / 1e9, g("SQ_INSTS_SALU") / 1e9, g("SQ_INSTS_BRANCH") / 1e9, g("SQ_INSTS_LDS") / 1e9, g("SQ_INSTS_VMEM_WR") / 1e6,
      g("SQ_WAVE_CYCLES") / 1e9, 100 * g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), 100 * g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
      100 * g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_LDS_IDX_ACTIVE") / 256e6, g("SQ_LDS_BANK_CONFLICT") / 256e6, g("SQ_LDS_ADDR_CONFLICT") / 256e6,
      g("SQ_INSTS_VALU") * 4 / (1024 * 2.38e9) * 1e3, (g("SQ_INSTS_SALU") + g("SQ_INSTS_BRANCH")) / (256 * 2.38e9) * 1e3))
PY
done
