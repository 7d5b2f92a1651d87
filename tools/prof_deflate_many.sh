#!/bin/bash
# per-kernel times of a call of N long streams deflated (tools/exp_deflate_many.py under rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-64x1048576}; do
  set -- ${cfg%x*} ${cfg#*x}
  d=gpurun_out/prof_defl_$1x$2; rm -rf $d; mkdir -p $d
  N=$1 LEN=$2 REPS=5 rocprofv3 --kernel-trace --stats -d $d -o many --output-format csv -- python3 tools/exp_deflate_many.py 2>/dev/null | grep -v "^[WE]2026"
  python3 - $d <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + '/many_kernel_stats.csv')))
print("  " + "  ".join("%s %.3f" % (r['Name'].split('(')[0].replace('zd::', '').replace('_kernel', ''), float(r['AverageNs']) / 1e6 * (int(r['Calls']) / 6.0)) for r in rows if r['Name'].startswith('zd::')))
PY
done
