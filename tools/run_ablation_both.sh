#!/bin/bash
# byte-layout and packed cross, full and without the mask role, through the C ABI
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for v in FULL NO_MASKROLE; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -w \
    -o gpurun_out/ablate/ab_$v tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc -DBH_ABLATE_$v -DBH_PACKED_R=${R:-2} -w \
    -o gpurun_out/ablate/abp_$v tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
  echo "== $v"
  ./gpurun_out/ablate/ab_$v 0; ./gpurun_out/ablate/ab_$v 1
  for a in "0 1 1" "0 0 1" "0 0 0" "0 1 0" "1 1 1"; do ./gpurun_out/ablate/abp_$v $a; done
done
