#!/bin/bash
# builds a libgsr_hip variant with flags applied to EVERY translation unit: tools/build_full_variant.sh name "-DFLAGS"
# -> tools/variants/libgsr_hip.<name>.so (tools/gpu_round6.sh swaps the variants in on the GPU box)
set -eu
cd "$(dirname "$0")/../gsworld_amd/csrc"
name=$1; extra=${2:-}
OUT=../../tools/variants; mkdir -p $OUT /tmp/gsr_variant_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-function"
for f in *.hip; do
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o /tmp/gsr_variant_$name/${f%.hip}.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgsr_hip.$name.so /tmp/gsr_variant_$name/*.o
echo built $OUT/libgsr_hip.$name.so
