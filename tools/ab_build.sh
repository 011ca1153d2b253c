#!/bin/bash
# A/B builds for experiments: a COPY of breedgym_amd/csrc with sed edits applied, compiled HERE (hipcc cross-compiles)
# into tools/_ab/<name>/ (git-ignored, travels to the GPU box) as libbreedhip.so + a C-ABI timing driver.  The shipped
# sources carry no experiment switches: a variant is a patch, named and kept here for as long as it is being measured.
# usage: tools/ab_build.sh <name> [--driver tools/ablate_gebv.cpp] [sed-expression ...]
#   e.g. tools/ab_build.sh slots4 's/kDmaSlots = 6/kDmaSlots = 4/'
set -e
cd "$(dirname "$0")/.."
name=$1; shift
drv=tools/ablate_packed.cpp
if [ "$1" = "--driver" ]; then drv=$2; shift 2; fi
out=tools/_ab/$name
rm -rf $out; mkdir -p $out/src
cp breedgym_amd/csrc/*.h breedgym_amd/csrc/*.hip $out/src/
for e in "$@"; do
  before=$(cat $out/src/*.h $out/src/*.hip | md5sum)
  sed -i -E "$e" $out/src/*.h $out/src/*.hip
  [ "$before" != "$(cat $out/src/*.h $out/src/*.hip | md5sum)" ] || { echo "sed expression changed nothing: $e"; exit 1; }
done
for u in breedhip cross gebv select; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Iinclude -w -c $out/src/$u.hip -o $out/$u.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libbreedhip.so $out/breedhip.o $out/cross.o $out/gebv.o $out/select.o
hipcc -O2 -std=c++17 -Iinclude -w -o $out/driver $drv -L$out -lbreedhip -Wl,-rpath,'$ORIGIN'
rm -f $out/*.o
echo built $out
