#!/bin/bash
# Builds tools/floor/libfloor.so (measurement aid; the .so is git-ignored and travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
[ libfloor.so -nt floor_kernels.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libfloor.so floor_kernels.hip
echo built tools/floor/libfloor.so
