#!/bin/bash
# byte-layout cross: full, without the mask role, without the row role (timing only)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for v in FULL NO_MASKROLE NO_ROWROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v $EXTRA -w \
    -o gpurun_out/ablate/ab_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  echo "== $v"
  ./gpurun_out/ablate/ab_$v 0; ./gpurun_out/ablate/ab_$v 1; ./gpurun_out/ablate/ab_$v 0 100000 4 2048
done
