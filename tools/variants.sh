#!/bin/bash
# Developer tool: build A/B variants of one kernel file into variants/lib_<name>.so
#   tools/variants.sh hist.hip base: flr:-DPPQHIP_V_FLR ...
# then on the GPU box:  PPQHIP_LIBRARY=variants/lib_flr.so python tools/microbench.py ...
set -e
cd "$(dirname "$0")/../ppq_amd/csrc"
make -s
src=$1; shift
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $defs -c $src -o build/var_$name.o
  objs=$(ls build/*.o | grep -v "build/var_" | grep -v "build/${src%.hip}.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib_$name.so $objs build/var_$name.o
  echo built variants/lib_$name.so "($defs)"
done
