#!/bin/bash
# Runs ON THE GPU BOX: the round-3 numbers DESIGN.md / profiles/ quote (trimmed collect_profiles.sh)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r3c}
rm -rf $O; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "quantile or to_int or lsq or minmax" -x) > $O/pytest_subset.log 2>&1
echo "rc=$?" >> $O/pytest_subset.log
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5"
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
cd $R
timeout 400 python tools/microbench.py --tensors A,B,Bx32 > $O/microbench_randn.txt 2>&1
timeout 300 python tools/microbench.py --tensors B,Bx32 --relu --only hist,minmax,quantile > $O/microbench_relu.txt 2>&1
timeout 300 python tools/multi_bench.py > $O/multi_bench.txt 2>&1
cd /tmp
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > /dev/null 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
tail -4 $O/pytest_subset.log; cut -c1-120 $O/microbench_randn.txt | grep -v amdgpu; cat $O/kernel_times_bench.txt; grep -E "lsq|finish" $O/kernel_times_micro.txt
