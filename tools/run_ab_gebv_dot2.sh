#!/bin/bash
# A/B: GEBV of the pipelined bit-plane stream as int16 digit dot products (default) or f64 FMAs
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_FULL -w -o gpurun_out/ablate/abg_dot2 tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_GEBV_F64 -DBH_ABLATE_FULL -w -o gpurun_out/ablate/abg_f64 tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
for rep in 1 2 3; do for k in dot2 f64; do
  echo -n "$k eager: "; ./gpurun_out/ablate/abg_$k 0 1 1
  echo -n "$k lazy: "; ./gpurun_out/ablate/abg_$k 0 0 1
done; done
