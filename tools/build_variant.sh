#!/bin/bash
# A variant build of libzipc_hip.so for same-box A/B runs (tools/ab_wall.sh):
#   tools/build_variant.sh NAME "-DZD_SCAN_ROUNDS=2 ..." [source ...]   -> zipc_amd/lib/libzipc_hip_NAME.so
# The named sources (default: deflate) are compiled again with the flags and linked with the product's
# other objects.  Variants are experiments: the tests and the bench only ever load libzipc_hip.so.
set -eu
NAME=${1:?name}; FLAGS=${2:-}; shift; shift || true
SRCS=${*:-deflate}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/zipc_amd/csrc
make -s -C "$C" -j4 all
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJS=""
for o in api inflate checksum deflate; do
  if [[ " $SRCS " == *" $o "* ]]; then
    $HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c "$C/$o.hip" -o "$C/build/${o}_$NAME.o"
    OBJS="$OBJS $C/build/${o}_$NAME.o"
  else
    OBJS="$OBJS $C/build/$o.o"
  fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/zipc_amd/lib/libzipc_hip_$NAME.so" $OBJS
echo "zipc_amd/lib/libzipc_hip_$NAME.so"
