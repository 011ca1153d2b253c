#!/bin/bash
# A/B: rows per wave in the byte-layout cross (same box, same run), full launch and row role alone
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for r in 2 4; do
  for v in FULL NO_MASKROLE; do
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ROWS_PER_WAVE_CROSS=$r -DBH_ABLATE_$v -w \
      -o gpurun_out/ablate/abr_${r}_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  done
done
for rep in 1 2; do for r in 2 4; do for v in FULL NO_MASKROLE; do
  echo -n "R=$r $v: "; ./gpurun_out/ablate/abr_${r}_$v 0
  echo -n "R=$r $v: "; ./gpurun_out/ablate/abr_${r}_$v 1
done; done; done
