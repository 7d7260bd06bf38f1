#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/quantile_profile.sh <tag>): parity tests of quantile / minmax / lsq, the per-tensor
# diagnosis of a real calibration (tools/quantile_diag.py), variants/lib_*.so built by tools/variants.sh, micro-benchmarks,
# rocprofv3 kernel traces of the micro-benchmark and of bench.py --method percentile (profiles/r03_quantile_*)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r3q3}
rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests -q -m gpu -k "quantile or minmax or lsq or percentile" -x) > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/quantile_diag.py 4 > $O/diag.txt 2>&1
for v in $(ls variants/lib_qf*.so 2>/dev/null); do
  n=$(basename $v .so); echo "== $n" >> $O/variants.txt
  PPQHIP_LIBRARY=$R/$v timeout 200 python tools/microbench.py --tensors Bx32 --only quantile 2>&1 | grep quantile >> $O/variants.txt
done
for v in $(ls variants/lib_mmc*.so); do
  n=$(basename $v .so); echo "== $n" >> $O/variants.txt
  PPQHIP_LIBRARY=$R/$v timeout 200 python tools/microbench.py --tensors Bx32 --only minmax_c 2>&1 | grep minmax >> $O/variants.txt
done
timeout 300 python tools/microbench.py --tensors B,Bx32 --only quantile,minmax,lsq > $O/micro_randn.txt 2>&1
timeout 300 python tools/microbench.py --tensors Bx32 --relu --only quantile > $O/micro_relu.txt 2>&1
timeout 300 python tools/multi_bench.py > $O/multi_bench.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o micro -- python $R/tools/microbench.py --tensors Bx32 --only "quantile_t (hinted),minmax_c,lsq_bwd_t" > /dev/null 2>&1
cd $R
python tools/kernel_times.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
timeout 400 python bench.py --method percentile --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 > $O/bench_percentile.json 2> $O/bench_percentile.err
cd /tmp
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace2 -o insitu -- python $R/bench.py --method percentile --no-cpu-baseline --no-cpu-ops --pmc 0 --variants 0 > /dev/null 2>&1
cd $R
python tools/kernel_times.py $(find $O/trace2 -name "*kernel_trace.csv" | head -1) quantile > $O/kernel_times_insitu.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
tail -5 $O/pytest.log; grep "^call" $O/diag.txt; cat $O/variants.txt | cut -c1-80; grep -E "quantile|minmax|lsq" $O/micro_randn.txt $O/micro_relu.txt $O/multi_bench.txt | cut -c1-140; cat $O/kernel_times_micro.txt; cat $O/kernel_times_insitu.txt; cut -c1-300 $O/bench_percentile.json
