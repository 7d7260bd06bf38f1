set -x
python -m pytest tests -m gpu -q -k "yolov6s or bias_correction or learned_step or lsq" > gpurun_out/r2_pytest7.log 2>&1; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r2_pytest7.log | tail -15
python tools/lsq_variants.py > gpurun_out/r2_lsq_variants.txt 2>&1; cat gpurun_out/r2_lsq_variants.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/r2_instcount -o ic -- python $GRAFT_REPO_ROOT/tools/hist_variants.py default.so > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_nofind -o nf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 --miopen-find 0 --repeats 1 > $GRAFT_REPO_ROOT/gpurun_out/r2_bench_nofind.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/r2_instcount $GRAFT_REPO_ROOT/gpurun_out/r2_nofind -name "*.db" -delete; find $GRAFT_REPO_ROOT/gpurun_out/r2_nofind -name "*kernel_trace.csv" -delete
ls -la $GRAFT_REPO_ROOT/gpurun_out/r2_instcount
