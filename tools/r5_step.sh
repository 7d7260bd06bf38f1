#!/bin/bash
# Runs ON THE GPU BOX: targeted parity tests of the round-5 changes, config-5 bench child, child repeat pattern, Bx32 A/B.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_step}
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py tests/test_gpu_finetune.py "tests/test_gpu_calibration.py::test_bench_two_ranks_on_one_gpu_configs_3_and_5" "tests/test_gpu_calibration.py::test_bench_two_ranks_on_one_gpu" -q -m gpu -x) > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
C="--warmup 1 --variants 0 --pmc 0 --no-cpu-baseline --no-cpu-ops --settle-ms 100 --miopen-find 1"
(time python bench.py --workload yolov6s_int4_lsq --steps 8 --batch 8 --repeats 3 $C) > $O/cfg5.json 2> $O/cfg5.err
tail -1 $O/cfg5.json | cut -c1-2500
tail -3 $O/cfg5.err
python bench.py --workload resnet50 --method percentile --steps 16 --batch 32 --repeats 4 $C 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('percentile values', j['values'], j['roofline']['frac'])"
python bench.py --workload vit_b16_fp8 --steps 8 --batch 16 --repeats 4 $C 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('vit values', j['values'], j['roofline']['frac'])"
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_bx32 -o floor -- python $R/tools/floor_table.py --tag bx32 --product-only --rounds 2 --iters 30 --shape 32,512,56,56 --libs r04=variants/lib_r04.so --cases hist_sym > $O/l2l_bx32.txt 2>&1
cd $R
python tools/floor_report.py "$(find $O/trace_bx32 -name '*kernel_trace.csv' | head -1)" gpurun_out/floor_manifest_bx32.json > $O/floor_bx32.txt 2>&1
rm -rf $O/trace_bx32
grep -v "^#" $O/floor_bx32.txt | cut -c3-75,92-
