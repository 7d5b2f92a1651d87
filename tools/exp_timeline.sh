#!/bin/bash
# One step's kernel timeline (start / end per launch, by queue) of the C2 shape: who runs beside whom.
#   gpurun -- 'bash tools/exp_timeline.sh'   -> gpurun_out/timeline/timeline.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/timeline; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DATA=${DATA:-c2} REPS=2 rocprofv3 --kernel-trace -d "$OUT/tr" -o t --output-format csv -- python3 "$ROOT/tools/exp_wall.py" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "zd::" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("zd::")[1].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
# the last deflate step: from the last deflate_offsets on
starts = [i for i, r in enumerate(rows) if r[2] == "deflate_offsets_kernel"]
i0 = starts[-1]
t0 = rows[i0][0]
with open(out + "/timeline.txt", "w") as w:
    for s, e, n, q in rows[i0:]:
        w.write("%9.3f %9.3f  %7.3f ms  q%s  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
print(open(out + "/timeline.txt").read())
PY
