#!/bin/bash
# Runs ON THE GPU BOX at the end of the round: the traces of the bench command and the micro-benchmark (tools/r6_trace.sh), the
# two-launch quantile's kernel medians per size, its first-call / zeroed-hint / settled timings, and the soaks.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6close}
rm -rf $O; mkdir -p $O
bash tools/r6_trace.sh ${1:-r6close}_trace > $O/trace_stdout.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/qtrace -o q -- python $R/tools/quantile_hot_bench.py 1,4,8,16,32 100 > $O/quantile_hot_bench.txt 2>&1
cd $R
python tools/kernel_times.py $(find $O/qtrace -name "*kernel_trace.csv" | head -1) 2>&1 | grep -E "quantile_hot" > $O/quantile_kernel_times.txt
rm -rf $O/qtrace
cat $O/quantile_kernel_times.txt
python tools/quantile_cold_ab.py 2>&1 | grep -v amdgpu | tee $O/quantile_cold_ab.txt
for seed in 1 2 3; do (timeout 600 python tools/quantile_soak.py 150 $seed single) 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/soak.txt; done
(timeout 600 python tools/quantile_soak.py 60 5) 2>&1 | tail -1 | tee -a $O/soak.txt
(timeout 600 python tools/quantile_soak.py 30 6 big) 2>&1 | tail -1 | tee -a $O/soak.txt
