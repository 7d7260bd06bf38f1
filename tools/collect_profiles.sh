#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): everything the numbers in DESIGN.md / profiles/ come from.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# PMC counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa tracing).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
cd $R
python tools/microbench.py --tensors A,B,Bx32 > $O/microbench_randn.txt 2>&1
python tools/microbench.py --tensors B,Bx32 --relu --only hist,minmax,quantile > $O/microbench_relu.txt 2>&1
python tools/multi_bench.py > $O/multi_bench.txt 2>&1
python bench.py --workload resnet50_cfg3 --pmc 0 > $O/bench_cfg3.json 2>/dev/null
python bench.py --workload vit_b16_fp8 --pmc 0 --batch 16 > $O/bench_cfg4.json 2>/dev/null
for b in 1 8; do python bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --batch $b --steps $((256 / b)) > $O/bench_batch$b.json 2>/dev/null; done
python -W ignore tools/reference_on_hip_bench.py 32 8 2>/dev/null | tail -1 > $O/reference_on_hip.txt
python -W ignore tools/reference_on_hip_bench.py 1 64 2>/dev/null | tail -1 >> $O/reference_on_hip.txt
python -m pytest tests/test_gpu_reference.py -m gpu -q -W ignore -v 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" > $O/reference_tests.txt
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > /dev/null 2>&1
cd $R
# keep the merge-back small: only the CSV summaries
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*_agent_info.csv" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O
tail -1 $O/bench.json | cut -c1-600
