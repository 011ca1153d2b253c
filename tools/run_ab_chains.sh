#!/bin/bash
# A/B: f64 accumulation chains per GEBV sum in the byte-layout cross (same box, same run)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for nc in 1 2 4; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_GEBV_CHAINS=$nc -w \
    -o gpurun_out/ablate/abc_$nc tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
done
for rep in 1 2 3; do for nc in 1 2 4; do echo -n "chains=$nc "; ./gpurun_out/ablate/abc_$nc 0; done; done
echo "--- 100k markers, 8192 crosses"
for nc in 1 4; do echo -n "chains=$nc "; ./gpurun_out/ablate/abc_$nc 0 100000 4 2048; done
