#!/bin/bash
# A/B at configs[2] size, shared keys (2048 mask units, 1024 row workgroups): mask units in turns or not
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for t in 1 4; do
  for v in FULL NO_MASKROLE; do
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_MASK_TURNS_MAX=$t -DBH_ABLATE_$v -w \
      -o gpurun_out/ablate/abt_${t}_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  done
done
for rep in 1 2 3; do for t in 1 4; do for v in FULL NO_MASKROLE; do
  echo -n "turns_max=$t $v: "; ./gpurun_out/ablate/abt_${t}_$v 0 100000 4 2048
done; done; done
