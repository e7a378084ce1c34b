#!/bin/bash
# Builds a variant of libsiftgpu.so with extra -D flags for one source file (tuning experiments):
#   profiles/build_variant.sh <name> <source.hip> "<flags>"   ->  build/variants/lib_<name>.so
# The other objects are taken as built in sift_amd/csrc.  Run a benchmark against it with SIFTGPU_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; flags=$3
out=$ROOT/build/variants
mkdir -p $out
cd $ROOT/sift_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $flags -c $src -o $out/${name}_${src%.hip}.o
objs=""
for f in siftgpu kernels_image kernels_blur kernels_detect kernels_describe kernels_match sg_comm; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $out/${name}_$f.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/lib_$name.so $objs sg_wire.o -ldl
echo built $out/lib_$name.so
