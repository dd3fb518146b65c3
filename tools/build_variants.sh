#!/bin/bash
# builds libgsr_hip variants that differ in ONE translation unit: tools/build_variants.sh <file.hip> name1:"-DFLAGS" name2:"-DFLAGS" ...
# outputs tools/variants/libgsr_hip.<name>.so (tools/gpu_variants.sh swaps them in on the GPU box)
set -eu
cd "$(dirname "$0")/../gsworld_amd/csrc"
SRC=$1; shift
OUT=../../tools/variants; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-function"
make -s -j8
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $extra -c $SRC -o /tmp/variant_$name.o
  objs=$(ls *.o | grep -v "^${SRC%.hip}.o$" | grep -v "_coopspin.o$\|_expacc.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgsr_hip.$name.so $objs /tmp/variant_$name.o
  echo built $OUT/libgsr_hip.$name.so
done
