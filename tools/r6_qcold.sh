#!/bin/bash
# Runs ON THE GPU BOX: the cold (exact passes) and hot timings of the two-launch quantile + its tests
set -u
export TMPDIR=/tmp
R=$PWD
(timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "quantile_single or quantile_hints or hip_graph") 2>&1 | tail -2
(timeout 600 python tools/quantile_soak.py 100 11 single) 2>&1 | tail -1
for mode in cold hot; do
for sz in 1 8 32; do
  cd /tmp; rm -rf /tmp/qc
  rocprofv3 --output-format csv --kernel-trace -d /tmp/qc -o q -- python $R/tools/quantile_hot_bench.py $sz 40 $mode > /tmp/qc.txt 2>&1
  cd $R
  echo "$mode x$sz: $(python tools/kernel_times.py $(find /tmp/qc -name '*kernel_trace.csv' | head -1) hot_select | cut -c54-110)"
done
done
