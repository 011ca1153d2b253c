#!/bin/bash
# A/B of the fused launches: pipelined row roles at equal priority (default) against the plain loop
# (-DBH_ROW_PLAIN), byte layout and bit planes, shared and independent keys, 10k and 100k markers
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ablate
for k in PIPE PLAIN; do
  X=""; [ $k = PLAIN ] && X="-DBH_ROW_PLAIN"
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc $X -DBH_ABLATE_FULL -w -o gpurun_out/ablate/abb_$k tools/ablate_cross.cpp breedgym_amd/csrc/breedhip.hip
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc $X -DBH_ABLATE_FULL -w -o gpurun_out/ablate/abq_$k tools/ablate_packed.cpp breedgym_amd/csrc/breedhip.hip
done
for rep in 1 2; do for k in PIPE PLAIN; do
  echo -n "$k bytes shared: "; ./gpurun_out/ablate/abb_$k 0
  echo -n "$k bits eager shared: "; ./gpurun_out/ablate/abq_$k 0 1 1
  echo -n "$k bits lazy shared: "; ./gpurun_out/ablate/abq_$k 0 0 1
done; done
for k in PIPE PLAIN; do
  echo -n "$k bytes independent: "; ./gpurun_out/ablate/abb_$k 1
  echo -n "$k bits eager independent: "; ./gpurun_out/ablate/abq_$k 1 1 1
  echo -n "$k bytes 100k shared: "; ./gpurun_out/ablate/abb_$k 0 100000 4 2048
  echo -n "$k bits eager 100k shared: "; ./gpurun_out/ablate/abq_$k 0 1 1 100000 4 2048
  echo -n "$k bits lazy 100k shared: "; ./gpurun_out/ablate/abq_$k 0 0 1 100000 4 2048
done
