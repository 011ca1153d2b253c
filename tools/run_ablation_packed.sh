#!/bin/bash
# bit-plane cross: full, without the mask role, without the row role (timing only): tools/run_ablation_packed.sh [M E n]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for v in FULL NO_MASKROLE NO_ROWROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -w \
    -o gpurun_out/ablate/abp_$v tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
  echo "== $v"
  for a in "0 1 1" "0 0 1" "0 1 0"; do ./gpurun_out/ablate/abp_$v $a "$@"; done
done
