#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the driver's bench command (the timed passes only: no variants, no children)
# and of the micro-benchmark; the summaries go to gpurun_out/r6trace (copy to profiles/r06_*).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6trace}
rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5"
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > $O/microbench_randn.txt 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
for d in trace micro_trace; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/trace $O/micro_trace
cat $O/kernel_times_bench.txt; tail -1 $O/bench_traced.json | cut -c1-300; grep -E "quantile|hist|minmax|fq_linear_c |channel_sum" $O/microbench_randn.txt | cut -c1-130
