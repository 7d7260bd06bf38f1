#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6qp}
rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o q -- python $R/tools/quantile_hot_bench.py ${2:-1,2,8,32} 100 > $O/run.txt 2>&1
cd $R
cat $O/run.txt | grep -v amdgpu.ids | cut -c1-250
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) quantile | tee $O/kernel_times.txt
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
