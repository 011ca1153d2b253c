#!/bin/bash
# A/B: rows per wave in the bit-plane fused launch (pipelined row role), eager and lazy observations
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for r in 1 2 4; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_PACKED_R=$r -DBH_ABLATE_FULL -w -o gpurun_out/ablate/abr_$r tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
done
for rep in 1 2; do for r in 1 2 4; do
  echo -n "R=$r eager: "; ./gpurun_out/ablate/abr_$r 0 1 1
  echo -n "R=$r lazy: "; ./gpurun_out/ablate/abr_$r 0 0 1
done; done
