#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6q4}
rm -rf $O; mkdir -p $O
PPQHIP_LIBRARY=$R/variants/lib_qhtime.so python tools/quantile_hot_stamps.py 1,8,32 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "quantile_single or quantile_hints" --durations=5) > $O/pytest_quantile.txt 2>&1
tail -4 $O/pytest_quantile.txt
for sz in 1 2 8 32; do
  cd /tmp
  timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$sz -o q -- python $R/tools/quantile_hot_bench.py $sz 100 > $O/run_$sz.txt 2>&1
  cd $R
  grep "^x" $O/run_$sz.txt | cut -c1-60
  python tools/kernel_times.py $(find $O/trace_$sz -name "*kernel_trace.csv" | head -1) quantile | tee -a $O/kernel_times.txt
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
