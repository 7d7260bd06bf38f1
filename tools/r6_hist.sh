#!/bin/bash
# Runs ON THE GPU BOX: single-tensor histogram variants on B x {sizes} (tools/hist_stream_bench.py on variants/lib_dev.so)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6h}
rm -rf $O; mkdir -p $O
NAMES="rows persistent|rows stream K=2|rows stream K=4|rows all-up-front K<=8|oneshot default|oneshot rows8|oneshot rows2|oneshot direct atomics|oneshot direct, grid 256|oneshot reduce launch (r5)"
for sz in ${2:-2 4 8 16 32}; do
  cd /tmp
  PPQHIP_LIBRARY=$R/variants/lib_dev.so timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$sz -o h -- python $R/tools/hist_stream_bench.py $sz 60 > $O/run_$sz.txt 2>&1
  cd $R
  echo "== x$sz   (checksums: $(grep -c counted $O/run_$sz.txt) cases, $(grep counted $O/run_$sz.txt | awk '{print $NF}' | sort -u | tr '\n' ' '))" | tee -a $O/report.txt
  python tools/hist_stream_report.py $(find $O/trace_$sz -name "*kernel_trace.csv" | head -1) "$NAMES" | tee -a $O/report.txt
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
