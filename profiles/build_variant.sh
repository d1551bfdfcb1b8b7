#!/bin/bash
# Development aid: build a variant of the HIP library with extra -D knobs into variants/<name>.so (untracked; travels to
# the GPU box with the snapshot).   bash profiles/build_variant.sh <name> [-DKNOB=value ...]
# Run it with   VARGENO_HIP_LIB=$PWD/variants/<name>.so python3 bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
C=vargeno_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DVG_LIB_BUILD_ID=\"variant-$name\" "$@" -c -o variants/$name.o $C/vargeno_hip.hip
[ -f $C/vg_sort.o ] || make -s -C $C $PWD/$C/vg_sort.o
make -s -C $C $PWD/$C/vg_hostpack.o $PWD/$C/vg_hostpack_avx2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o variants/$name.so variants/$name.o $C/vg_sort.o $C/vg_hostpack.o $C/vg_hostpack_avx2.o -ldl -lpthread
rm -f variants/$name.o
echo built variants/$name.so
