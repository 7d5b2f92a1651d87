#!/bin/bash
# same-box A/B of builds of the checksum kernels on C3 (tools/exp_c3.py): exp_c3_ab.sh "<lib> <lib> ..."
for lib in $1; do echo $lib; ZIPC_HIP_LIB=$PWD/zipc_amd/lib/$lib python3 tools/exp_c3.py | cut -c1-330; done
