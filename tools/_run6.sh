set -x
python -m pytest tests -m gpu -q > gpurun_out/r2_pytest6.log 2>&1; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r2_pytest6.log | tail -25
