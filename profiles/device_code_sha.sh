#!/bin/bash
# sha256 of the machine code of every kernel in a built libvargeno_hip.so (the gfx950 code object's .text), with the library's build id:
# two builds that print the same line run the same kernels, whatever host code changed between them.
#   bash profiles/device_code_sha.sh [path/to/libvargeno_hip.so]
L=${1:-$(dirname $0)/../vargeno_amd/csrc/libvargeno_hip.so}
T=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin $L $T/discard.so
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb.bin --output=$T/co.o --unbundle
$B/llvm-objcopy --dump-section .text=$T/text.bin $T/co.o $T/discard.o
ID=$(python3 -c "import ctypes,sys; l=ctypes.CDLL('$L'); l.vg_build_id.restype=ctypes.c_char_p; print(l.vg_build_id().decode())")
echo "build $ID  gfx950 .text $(stat -c %s $T/text.bin) bytes  sha256 $(sha256sum $T/text.bin | cut -d' ' -f1)"
rm -rf $T
