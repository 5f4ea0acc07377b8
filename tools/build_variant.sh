#!/bin/bash
# Build an A/B variant of libvicasplat_hip.so: bash tools/build_variant.sh <tag> <file.hip (replacement source)> <name of the object it replaces, e.g. raster_bwd>
# -> variants/libvicasplat_hip_<tag>.so (git-ignored; travels with gpurun).  Use: VICASPLAT_HIP_LIB=variants/libvicasplat_hip_<tag>.so python ...
set -e
tag=$1; src=$2; obj=$3
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/variants
cd $R/vicasplat_amd/csrc
extra=""; [[ $obj == attention ]] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $extra -I$R/vicasplat_amd/csrc -c $src -o /tmp/variant_${tag}_$obj.o
objs=$(ls *.o | grep -v "^$obj.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libvicasplat_hip_$tag.so $objs /tmp/variant_${tag}_$obj.o
echo built $R/variants/libvicasplat_hip_$tag.so
