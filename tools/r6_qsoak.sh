#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6qs}
rm -rf $O; mkdir -p $O
for seed in 1 2 3; do (timeout 600 python tools/quantile_soak.py 150 $seed single) 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/soak_single.txt; done
(timeout 600 python tools/quantile_soak.py 60 5) 2>&1 | tail -2 | tee -a $O/soak.txt
(timeout 600 python tools/quantile_soak.py 30 6 big) 2>&1 | tail -2 | tee -a $O/soak.txt
(timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "quantile_single") 2>&1 | tail -3 | tee $O/pytest.txt
