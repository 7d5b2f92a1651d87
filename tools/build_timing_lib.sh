#!/bin/bash
# Builds a timing-only variant of libzipc_hip.so for the phase tools:
#   tools/build_timing_lib.sh ZD_MATCH_PHASES   [out.so]   -> tools/exp_match_phases.py
#   tools/build_timing_lib.sh ZD_MATCH_COUNTS   [out.so]   -> tools/exp_wall.py with MATCH_COUNTS=1 (what the pools' waves do, counted)
#   tools/build_timing_lib.sh ZD_EMIT_PHASES    [out.so]   -> tools/exp_emit_phases.py
#   tools/build_timing_lib.sh ZD_INFLATE_PHASES [out.so]   -> tools/exp_inflate_phases.py
# then run the tool with ZIPC_HIP_LIB=<out.so>.  The product library is not touched: the one
# source that carries the macro is compiled again with -D<MACRO> and linked with the product's
# other objects (make -C zipc_amd/csrc first).  Such a build returns clock stamps where the
# product returns results, or exports a debug entry point: never ship or test parity with it.
set -eu
MACRO=${1:?ZD_MATCH_PHASES | ZD_MATCH_COUNTS | ZD_PARSE_COUNTS | ZD_PARSE_PHASES | ZD_EMIT_PHASES | ZD_INFLATE_PHASES}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/zipc_amd/csrc
OUT=${2:-$C/build/timing_$MACRO.so}
case $MACRO in ZD_INFLATE_PHASES) SRC=inflate;; *) SRC=deflate;; esac
make -s -C "$C" -j4 all
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -D$MACRO -c "$C/$SRC.hip" -o "$C/build/${SRC}_$MACRO.o"
OBJS=""; for o in api inflate checksum deflate; do if [ $o = $SRC ]; then OBJS="$OBJS $C/build/${SRC}_$MACRO.o"; else OBJS="$OBJS $C/build/$o.o"; fi; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "$OUT"
