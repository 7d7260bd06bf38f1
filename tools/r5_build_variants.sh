#!/bin/bash
# Developer tool (round 5): the A/B libraries tools/r5_floor.sh measures on the GPU box.
#   variants/lib_r04.so    : the kernels as committed at the end of round 4 (3e97a8c) -- the "before" of every comparison
#   variants/lib_dev.so    : HEAD + -DPPQHIP_DEV_KNOBS (env knobs: PPQHIP_DEV_HIST_WG / _SMALL / _ATOMIC, swept by tools/floor_table.py)
#   variants/lib_<name>.so : one source file rebuilt with other -D flags (arguments, see the end of this file)
set -e
cd "$(dirname "$0")/.."
R=$PWD
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
make -s -C ppq_amd/csrc
mkdir -p variants ppq_amd/csrc/build/r04/ppq_amd/csrc ppq_amd/csrc/build/r04/include
if [ ! -f variants/lib_r04.so ]; then
  B=ppq_amd/csrc/build/r04
  for f in common.hpp runtime.hip linear.hip floating.hip hist.hip reduce.hip quantile.hip search.hip train.hip; do git show 3e97a8c:ppq_amd/csrc/$f > $B/ppq_amd/csrc/$f; done
  git show 3e97a8c:include/ppq_hip.h > $B/include/ppq_hip.h
  objs=""
  for f in runtime linear floating hist reduce quantile search train; do $HIPCC $FLAGS -c $B/ppq_amd/csrc/$f.hip -o $B/$f.o & objs="$objs $B/$f.o"; done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o variants/lib_r04.so $objs
  echo built variants/lib_r04.so
fi
cd ppq_amd/csrc
build() {  # name src defs...
  name=$1; src=$2; shift 2
  stem=${src%.hip}
  $HIPCC $FLAGS "$@" -c $src -o build/var_${name}.o
  others=$(ls build/*.o | grep -v "build/var_" | grep -v "build/$stem.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/variants/lib_$name.so $others build/var_${name}.o
  echo "built variants/lib_$name.so ($*)"
}
build dev hist.hip -DPPQHIP_DEV_KNOBS &
# further A/B builds: name:file:"flags", e.g.  tools/r5_build_variants.sh fqS2:linear.hip:"-DPPQHIP_FQ_SMALL_U=2" fqnorcp:linear.hip:"-DPPQHIP_FQ_RCP=0"
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; file=${rest%%:*}; defs=${rest#*:}
  build $name $file $defs &
done
wait
