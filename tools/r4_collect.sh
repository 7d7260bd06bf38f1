#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-4 numbers DESIGN.md / profiles/ quote.
#   gpurun --timeout 2400 -- 'bash tools/r4_collect.sh'
# PMC counters are collected by bench.py itself in separate --pmc child passes (kernel tracing only next to them).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r4}
rm -rf $O; mkdir -p $O
# 1. the driver's command (headline + variants + live PMC + cpu baseline)
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
# 2. the same command under rocprofv3 (no PMC, no variants): per-kernel average durations
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5"
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
# 3. config 5 under rocprofv3: the multi-tensor LSQ backward in situ
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace5 -o cfg5 -- python $R/bench.py --workload yolov6s_int4_lsq --steps 8 --batch 8 \
    --warmup 1 --repeats 1 --variants 0 --pmc 0 --no-cpu-baseline --no-cpu-ops --settle-ms 0 --miopen-find 0 > $O/bench_cfg5_traced.json 2>/dev/null
cd $R
# 4. single-tensor microbench (rotating buffers), multi-tensor bench, facade host time
timeout 400 python tools/microbench.py --tensors A,B,Bx32 > $O/microbench_randn.txt 2>&1
timeout 300 python tools/microbench.py --tensors B,Bx32 --relu --only hist,minmax,quantile > $O/microbench_relu.txt 2>&1
timeout 300 python tools/multi_bench.py > $O/multi_bench.txt 2>&1
timeout 120 python tools/call_overhead.py > $O/call_overhead.txt 2>&1
# 5. A/B variants built by tools/variants.sh (if present)
for v in variants/lib_*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so)
  PPQHIP_LIBRARY=$R/$v timeout 200 python tools/microbench.py --tensors Bx32 --only to_int,fq_linear > $O/variant_${n}_micro.txt 2>&1
  PPQHIP_LIBRARY=$R/$v timeout 200 python tools/multi_bench.py 2>&1 | grep -E "weights|fq_linear|minmax_c" > $O/variant_${n}_multi.txt
done
# 6. per-kernel medians of the microbench under rocprofv3
cd /tmp
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > /dev/null 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
python tools/kernel_times.py $(find $O/trace5 -name "*kernel_trace.csv" | head -1) > $O/kernel_times_cfg5.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O
tail -1 $O/bench.json | cut -c1-900; grep -v amdgpu $O/microbench_randn.txt | cut -c1-130; cat $O/multi_bench.txt | grep -v amdgpu; cat $O/variant_*; grep -v amdgpu $O/call_overhead.txt
