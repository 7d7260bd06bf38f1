set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest1.log 2>&1; tail -5 gpurun_out/r2_pytest1.log
python tools/hist_variants.py > gpurun_out/r2_variants1.log 2>&1; tail -80 gpurun_out/r2_variants1.log
