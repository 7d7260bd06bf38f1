#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_step2}
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py tests/test_gpu_fp8_reference.py tests/test_gpu_reference.py -q -m gpu -x) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_ab -o floor -- python $R/tools/floor_table.py --tag ab2 --rounds 3 --libs r04=variants/lib_r04.so --cases fq_linear > $O/l2l_ab.txt 2>&1
cd $R
python tools/floor_report.py "$(find $O/trace_ab -name '*kernel_trace.csv' | head -1)" gpurun_out/floor_manifest_ab2.json > $O/floor_ab.txt 2>&1
rm -rf $O/trace_ab
grep -v "^#" $O/floor_ab.txt | grep "lib\[\|floor_copy\|floor_empty grid=256 block=256\|floor_read block=256 U=2 grid=one" | cut -c3-75,92-
C="--warmup 1 --variants 0 --pmc 0 --no-cpu-baseline --no-cpu-ops --settle-ms 100 --miopen-find 1"
python bench.py --workload yolov6s_int4_lsq --steps 8 --batch 8 --repeats 3 $C 2>$O/cfg5.err | tail -1 > $O/cfg5.json
python -c "import json; j=json.load(open('$O/cfg5.json')); print(j['value'], j['values'], j['config']['step'])"
