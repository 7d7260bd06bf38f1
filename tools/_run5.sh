set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest5.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest5.log | tail -15
python tools/microbench.py --tensors A,B,Bx32 > gpurun_out/r2_micro_randn.txt 2>&1; cat gpurun_out/r2_micro_randn.txt
python tools/microbench.py --tensors B,Bx32 --relu --only hist,minmax,quantile > gpurun_out/r2_micro_relu.txt 2>&1; cat gpurun_out/r2_micro_relu.txt
python tools/multi_bench.py > gpurun_out/r2_multi_bench.txt 2>&1; cat gpurun_out/r2_multi_bench.txt
