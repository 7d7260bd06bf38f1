#!/bin/bash
# Developer tool: A/B builds of hist.hip (tile geometry, EXEC-mask commits, ...) into variants/lib_<name>.so,
# plus round 1's kernel (commit af018bf) as the baseline.  Measured by tools/hist_variants.py.
#   tools/build_hist_variants.sh name:"-Dflags" ...        (no arguments: the default set)
set -e
cd "$(dirname "$0")/../ppq_amd/csrc"
make -s
mkdir -p ../../variants build/var
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
others=$(ls build/*.o | grep -v "build/hist.o")
build() {  # name defs...
  name=$1; shift
  $HIPCC $FLAGS "$@" -c hist.hip -o build/var/hist_$name.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $others build/var/hist_$name.o
  echo "built variants/lib_$name.so ($*)"
}
if [ ! -f ../../variants/lib_r01.so ]; then
  mkdir -p build/r01/ppq_amd/csrc build/r01/include
  for f in common.hpp hist.hip; do git show af018bf:ppq_amd/csrc/$f > build/r01/ppq_amd/csrc/$f; done
  git show af018bf:include/ppq_hip.h > build/r01/include/ppq_hip.h
  $HIPCC $FLAGS -c build/r01/ppq_amd/csrc/hist.hip -o build/var/hist_r01.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_r01.so $others build/var/hist_r01.o
  echo "built variants/lib_r01.so (round-1 hist.hip)"
fi
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  [ "$defs" == "$name" ] && defs=""
  build $name $defs
done
