#!/bin/bash
# Build one variant of the library + the C++ ablation driver HERE (hipcc cross-compiles), into tools/_ab/
# (git-ignored, travels to the GPU box).  usage: tools/ab_build.sh <name> [-DMACRO ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
name=$1; shift
drv=tools/ablate_packed.cpp
if [ "$1" = "--gebv" ]; then drv=tools/ablate_gebv.cpp; shift; fi
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ibreedgym_amd/csrc "$@" -w -o tools/_ab/$name $drv breedgym_amd/csrc/breedhip.hip
echo built tools/_ab/$name
