set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest4.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest4.log | tail -15
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_btrace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cpu-ops --pmc 0 > $GRAFT_REPO_ROOT/gpurun_out/r2_bench4_traced.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/r2_btrace -name "*.db" -delete; find $GRAFT_REPO_ROOT/gpurun_out/r2_btrace -name "*kernel_trace.csv" -delete
