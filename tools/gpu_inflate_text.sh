#!/bin/bash
# Phase clocks of the timing build on text: the batch of 64 KiB streams and one 1 MiB stream.
cd "${GRAFT_REPO_ROOT:-.}"
export ZIPC_HIP_PHASE_LIB=zipc_amd/csrc/build/timing_ZD_INFLATE_PHASES.so
DOC=1 timeout 300 python tools/exp_inflate_phases.py 2>&1 | tail -2
DOC=1 N_STREAMS=2 LEN=1048576 timeout 300 python tools/exp_inflate_phases.py 2>&1 | tail -2
