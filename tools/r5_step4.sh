#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_step4}
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py "tests/test_gpu_calibration.py::test_multi_tensor_quantile_equals_single" -q -m gpu -x -k "quantile or Quantile or percentile") > $O/pytest.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -2
timeout 300 python tools/quantile_soak.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors A,B,Bx32 --only quantile > $O/micro.txt 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
rm -rf $O/micro_trace
grep "quantile" $O/micro.txt | grep -v rocprof; cat $O/kernel_times_micro.txt
python -c "
import bench, json
print(json.dumps(bench.north_star_b(2048)))"
