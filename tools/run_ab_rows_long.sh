#!/bin/bash
# A/B at configs[2] size (8192 crosses x 100k markers): rows per wave 1 vs 2, full launch and row role alone
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for r in 1 2; do
  for v in FULL NO_MASKROLE; do
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ROWS_PER_WAVE_CROSS=$r -DBH_ABLATE_$v -w \
      -o gpurun_out/ablate/abl_${r}_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  done
done
for rep in 1 2; do for r in 1 2; do for v in FULL NO_MASKROLE; do
  echo -n "R=$r $v: "; ./gpurun_out/ablate/abl_${r}_$v 0 100000 4 2048
done; done; done
for r in 1 2; do echo -n "R=$r FULL independent: "; ./gpurun_out/ablate/abl_${r}_FULL 1 100000 4 2048; done
