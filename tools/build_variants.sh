#!/bin/bash
# Builds measurement variants of the library HERE (hipcc cross-compiles): one per line "name flags..." of the given file
# -> tools/variants/libosm_<name>.so (git-ignored, travels with gpurun).  Run tools/run_variants.sh on the GPU box.
V=$(readlink -f "$1"); cd "$(dirname "$0")/../osmosis_diffusion_code_amd/csrc"
mkdir -p ../../tools/variants
while read -r name flags; do
  [ -z "$name" ] && continue
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize $flags -c igemm.hip -o /tmp/igemm_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm_$name.o norm.o elementwise.o guidance.o attention.o flash.o igemm_h.o norm_h.o elementwise_h.o -o ../../tools/variants/libosm_$name.so && echo "built $name" ) &
done < "$V"
wait
