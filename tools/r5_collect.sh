#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-5 numbers DESIGN.md / profiles/ quote.
#   gpurun --timeout 3000 -- 'bash tools/r5_collect.sh [tag]'
# PMC counters are collected by bench.py itself in separate --pmc child passes (kernel tracing only next to them).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5}
rm -rf $O; mkdir -p $O
rm -f $R/gpurun_out/parity_residue.jsonl
# 0. the whole GPU suite (also leaves gpurun_out/parity_residue.jsonl)
(time timeout 1500 python -m pytest tests -q -m gpu) > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_residue.jsonl $O/ 2>/dev/null
# 1. the driver's command (headline + variants + live PMC + cpu baseline)
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
# 2. the same command under rocprofv3 (no PMC, no variants): per-kernel average durations
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5"
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
# 3. config 5 under rocprofv3: kernels per replayed step
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace5 -o cfg5 -- python $R/bench.py --workload yolov6s_int4_lsq --steps 8 --batch 8 \
    --warmup 1 --repeats 1 --variants 0 --pmc 0 --no-cpu-baseline --no-cpu-ops --settle-ms 0 --miopen-find 1 > $O/bench_cfg5_traced.json 2>/dev/null
cd $R
# 4. the floor table on B (floor kernels + the library at HEAD + round 4's library, interleaved, 3 rounds) and on A / Bx8 / Bx32
run() {  # tag, args...
  tag=$1; shift
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$tag -o floor -- python $R/tools/floor_table.py --tag $tag "$@" > $O/l2l_$tag.txt 2>&1
  cd $R
  python tools/floor_report.py "$(find $O/trace_$tag -name '*kernel_trace.csv' | head -1)" gpurun_out/floor_manifest_$tag.json > $O/floor_$tag.txt 2>&1
  rm -rf $O/trace_$tag
}
L=""; [ -f variants/lib_r04.so ] && L="--libs r04=variants/lib_r04.so"
run b --rounds 3 $L
run a --product-only --rounds 2 --shape 1,3,224,224 $L
run bx8 --product-only --rounds 2 --iters 60 --shape 8,512,56,56 $L
run bx32 --product-only --rounds 2 --iters 30 --shape 32,512,56,56 $L
# 5. single-tensor microbench (rotating buffers), multi-tensor bench
timeout 400 python tools/microbench.py --tensors A,B,Bx32 > $O/microbench_randn.txt 2>&1
timeout 300 python tools/microbench.py --tensors B,Bx32 --relu --only hist,minmax,quantile > $O/microbench_relu.txt 2>&1
timeout 300 python tools/multi_bench.py > $O/multi_bench.txt 2>&1
# 6. per-kernel medians under rocprofv3
cd /tmp
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > /dev/null 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
python tools/kernel_times.py $(find $O/trace5 -name "*kernel_trace.csv" | head -1) > $O/kernel_times_cfg5.txt 2>&1
for d in trace trace5 micro_trace; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
rm -rf $O/trace $O/trace5 $O/micro_trace
du -sh $O
tail -1 $O/bench.json | cut -c1-1200; grep -v "^#" $O/floor_b.txt | cut -c3-75,92- | head -80
