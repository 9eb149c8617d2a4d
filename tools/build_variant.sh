#!/bin/bash
# usage: [VARIANT_FILE=mlp] tools/build_variant.sh <name> [-DSWITCH=value ...]   -> ab/<name>.so : the library with ONE source file
#        (default render.hip) rebuilt with extra defines, the other objects taken from the tree (same-box A/B via SN_LIB)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/ab/$name; mkdir -p $out
cd $root/sanerf-hq_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function"
vf=${VARIANT_FILE:-render}
for f in grid grid_binned encoders raymarch render heads mlp mlp_small linear optim; do
  [ "$f" = "$vf" ] && continue
  [ -f $root/sanerf-hq_amd/csrc/$f.o ] && cp $root/sanerf-hq_amd/csrc/$f.o $out/$f.o || /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $out/$f.o
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c $vf.hip -o $out/$vf.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $root/ab/$name.so
rm -rf $out
echo built ab/$name.so
