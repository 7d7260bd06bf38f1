#!/bin/bash
# Runs ON THE GPU BOX: the lean end-of-round verification at HEAD -- GPU suite, smoke(), the driver's bench command, its rocprofv3
# trace, the floor table on B (HEAD vs round 4 interleaved) and the microbench kernel medians.  (tools/r5_collect.sh = the full set.)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_final}
rm -rf $O; mkdir -p $O
rm -f $R/gpurun_out/parity_residue.jsonl
(time timeout 1500 python -m pytest tests -q -m gpu) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
cp $R/gpurun_out/parity_residue.jsonl $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
B="python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5"
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- $B > $O/bench_traced.json 2>/dev/null
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_b -o floor -- python $R/tools/floor_table.py --tag b --rounds 3 --libs r04=variants/lib_r04.so > $O/l2l_b.txt 2>&1
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 > $O/microbench_randn.txt 2>&1
cd $R
python tools/floor_report.py "$(find $O/trace_b -name '*kernel_trace.csv' | head -1)" gpurun_out/floor_manifest_b.json > $O/floor_b.txt 2>&1
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
for d in trace micro_trace; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/trace $O/trace_b $O/micro_trace
tail -1 $O/bench.json | cut -c1-600; grep "lib\[" $O/floor_b.txt | cut -c3-75,92-
