#!/bin/bash
# Developer tool: A/B builds of ONE kernel file (default hist.hip; SRC=linear.hip ... for another) into
# variants/lib_<name>.so, plus round 1's histogram kernel (commit af018bf) as the baseline.
# Measured by tools/hist_variants.py / tools/lsq_variants.py.
#   [SRC=linear.hip] tools/build_hist_variants.sh name:"-Dflags" ...
set -e
cd "$(dirname "$0")/../ppq_amd/csrc"
make -s
mkdir -p ../../variants build/var
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
SRC=${SRC:-hist.hip}
stem=${SRC%.hip}
others=$(ls build/*.o | grep -v "build/$stem.o")
build() {  # name defs...
  name=$1; shift
  $HIPCC $FLAGS "$@" -c $SRC -o build/var/${stem}_$name.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $others build/var/${stem}_$name.o
  echo "built variants/lib_$name.so ($*)"
}
if [ "$SRC" == "hist.hip" ] && [ ! -f ../../variants/lib_r01.so ]; then
  mkdir -p build/r01/ppq_amd/csrc build/r01/include
  for f in common.hpp hist.hip; do git show af018bf:ppq_amd/csrc/$f > build/r01/ppq_amd/csrc/$f; done
  git show af018bf:include/ppq_hip.h > build/r01/include/ppq_hip.h
  $HIPCC $FLAGS -c build/r01/ppq_amd/csrc/hist.hip -o build/var/hist_r01.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_r01.so $(ls build/*.o | grep -v "build/hist.o") build/var/hist_r01.o
  echo "built variants/lib_r01.so (round-1 hist.hip)"
fi
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  [ "$defs" == "$name" ] && defs=""
  build $name $defs
done
