#!/bin/bash
# A variant of libzipc_hip.so whose DEVICE code went through a rewrite of its assembly (an experiment's tool, like
# build_variant.sh; the product is never built this way):
#   tools/build_asm_variant.sh NAME 'sed-script' [source ...]     -> zipc_amd/lib/libzipc_hip_NAME.so
# e.g. the selects that read VCC in the VOP2 encoding as VOP3 (profiles/r06_vcc_select.txt):
#   tools/build_asm_variant.sh e64 's/v_cndmask_b32_e32 \(.*\), vcc$/v_cndmask_b32_e64 \1, vcc/'
# Steps per source: hipcc --cuda-device-only -S, sed, assemble, link the code object, bundle it, compile the host side with
# that bundle (-fcuda-include-gpubinary) -- what hipcc does in one go, with the sed in the middle.
set -eu
NAME=${1:?name}; SED=${2:?sed script}; shift; shift
SRCS=${*:-deflate inflate}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/zipc_amd/csrc; B=$C/build
make -s -C "$C" -j4 all
LLVM=/opt/rocm/lib/llvm/bin; HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
OBJS=""
for o in api inflate checksum deflate; do
  if [[ " $SRCS " == *" $o "* ]]; then
    $HIPCC $FLAGS --cuda-device-only -S "$C/$o.hip" -o "$B/${o}_$NAME.s" 2>/dev/null
    sed -i "$SED" "$B/${o}_$NAME.s"
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$B/${o}_$NAME.s" -o "$B/${o}_$NAME.dev.o"
    $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B/${o}_$NAME.co" "$B/${o}_$NAME.dev.o"
    $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
        -input=/dev/null -input="$B/${o}_$NAME.co" -output="$B/${o}_$NAME.hipfb"
    $HIPCC $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B/${o}_$NAME.hipfb" -c "$C/$o.hip" -o "$B/${o}_$NAME.o"
    OBJS="$OBJS $B/${o}_$NAME.o"
  else
    OBJS="$OBJS $B/$o.o"
  fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/zipc_amd/lib/libzipc_hip_$NAME.so" $OBJS
echo "zipc_amd/lib/libzipc_hip_$NAME.so"
