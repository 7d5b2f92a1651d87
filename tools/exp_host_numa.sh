#!/bin/bash
# Do the many-stream host forms care which NUMA node the calling process sits on?  The box's topology, the GPU's
# node, then tools/bench_host_forms.py pinned (taskset, before anything touches the GPU) to each node's cores in turn.
# usage (on the GPU box): bash tools/exp_host_numa.sh <tag>
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG/host_numa.txt; : > $O
{ lscpu | grep -i "numa\|model name\|^CPU(s)\|socket"; nproc; cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list"; } 2>&1 | tee -a $O
for d in /sys/bus/pci/devices/*; do
  if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] && [ "$(cat $d/class 2>/dev/null | cut -c1-6)" != "0x0604" ]; then
    echo "pci $(basename $d) class $(cat $d/class) numa_node $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null)"; fi
done 2>&1 | tee -a $O
python3 -c "
import torch
p = torch.cuda.get_device_properties(0)
print('device 0:', p.name, 'pci', getattr(p, 'pci_bus_id', None), getattr(p, 'pci_device_id', None), getattr(p, 'pci_domain_id', None))
" 2>&1 | tail -1 | tee -a $O
row() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); m=lambda x: sorted(x)[len(x)//2]; print(d["round_trip_ok"], "def median %.2f min %.2f max %.2f" % (m(d["deflate_ms_all"]), min(d["deflate_ms_all"]), max(d["deflate_ms_all"])), "| inf median %.2f min %.2f max %.2f" % (m(d["inflate_ms_all"]), min(d["inflate_ms_all"]), max(d["inflate_ms_all"])))'; }
N=${N_STREAMS:-4096}
for rep in 1 2; do
  for node in /sys/devices/system/node/node*; do
    cpus=$(cat $node/cpulist)
    echo "$(basename $node) cpus $cpus: $(N_STREAMS=$N REPS=7 taskset -c $cpus python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
  done
  echo "unpinned: $(N_STREAMS=$N REPS=7 python3 tools/bench_host_forms.py 2>/dev/null | row)" | tee -a $O
done
