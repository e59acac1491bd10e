#!/bin/bash
# Build a compile-time variant of the library for A/B runs: tools/build_variant.sh NAME file.hip -DFLAG=.. [...]
# -> infodiffusion_amd/variants/libinfodiff_hip_NAME.so (select with IDF_LIB=<path>); the other objects come from
# the normal in-tree build (infodiffusion_amd/build/*.o).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/infodiffusion_amd/variants
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c $root/infodiffusion_amd/csrc/$base.hip -o /tmp/${base}_$name.o
objs=$(ls $root/infodiffusion_amd/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/infodiffusion_amd/variants/libinfodiff_hip_$name.so $objs /tmp/${base}_$name.o
echo built $root/infodiffusion_amd/variants/libinfodiff_hip_$name.so
