#!/bin/bash
# Build a variant of the library with extra flags on ONE translation unit:  tools/build_variant.sh NAME FILE.hip "-DFLAG ..."
# -> gpurun_in/lib/libpalu_hip_NAME.so  (select with PALU_HIP_LIB=...; the other objects come from palu_amd/lib)
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; flags=$3
mkdir -p gpurun_in/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $flags -c palu_amd/csrc/$file -o gpurun_in/lib/$name.$file.o
objs=""
for o in palu_amd/lib/*.hip.o; do
  if [ "$(basename $o)" != "$file.o" ]; then objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs gpurun_in/lib/$name.$file.o -o gpurun_in/lib/libpalu_hip_$name.so
echo gpurun_in/lib/libpalu_hip_$name.so
