set -x
python -m pytest tests -m gpu -q -k "yolov6s or bias_correction or learned_step or lsq" > gpurun_out/r2_pytest11.log 2>&1; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r2_pytest11.log | tail -15
python tools/lsq_variants.py > gpurun_out/r2_lsq_variants2.txt 2>&1; cat gpurun_out/r2_lsq_variants2.txt
