#!/bin/bash
# developer: build libomnimamba_hip_<tag>.so that differs from the product library only in ONE source compiled with extra flags
# usage: tools/build_variant.sh <tag> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
OBJ=omnimamba_amd/lib/obj
mkdir -p omnimamba_amd/lib/obj_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -ffp-contract=fast "$@" -c omnimamba_amd/csrc/$SRC -o omnimamba_amd/lib/obj_var/${SRC%.hip}_$TAG.o
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic $OBJS omnimamba_amd/lib/obj_var/${SRC%.hip}_$TAG.o -o omnimamba_amd/lib/libomnimamba_hip_$TAG.so
echo omnimamba_amd/lib/libomnimamba_hip_$TAG.so
