#!/bin/bash
# Runs ON THE GPU BOX: the driver's three checks (GPU suite, smoke, bench) + the rocprofv3 summary of the bench command.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r4final}
rm -rf $O; mkdir -p $O
(time python -m pytest tests -x -q -m gpu) > $O/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 --steps 20 --warmup 5 > $O/bench_traced.json 2>/dev/null
cd $R
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_bench.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
tail -3 $O/pytest_gpu.log; tail -3 $O/smoke.log; tail -1 $O/bench.json | cut -c1-700; grep -E "hist_persistent|minmax_persistent" $O/kernel_times_bench.txt
