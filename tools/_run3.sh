set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest3.log 2>&1; tail -15 gpurun_out/r2_pytest3.log
python tools/run_reference_tests.py > gpurun_out/r2_reference_tests.txt 2>&1; tail -5 gpurun_out/r2_reference_tests.txt
( time python bench.py ) > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err; tail -c 3000 gpurun_out/r2_bench3.json; tail -5 gpurun_out/r2_bench3.err
