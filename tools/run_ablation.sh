#!/bin/bash
# builds and runs the ablation variants on the GPU box: tools/run_ablation.sh [M E n]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for v in FULL NO_ROWROLE NO_MASKROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -Wno-unused-result \
    -o gpurun_out/ablate/ab_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  echo "== $v"; ./gpurun_out/ablate/ab_$v "$@"
done
