#!/bin/bash
# builds and runs the ablation variants of the packed cross on the GPU box: rows per wave x roles
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for R in ${RS:-1 2 4}; do
for v in FULL NO_MASKROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -DBH_PACKED_R=$R -w \
    -o gpurun_out/ablate/abp_$v tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
  echo "== R=$R $v"
  for a in "0 1 1" "0 0 1" "0 0 0" "0 1 0"; do ./gpurun_out/ablate/abp_$v $a; done
done
done
