#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_step3}
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py tests/test_gpu_reference.py tests/test_gpu_finetune.py -q -m gpu -x) > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
C="--warmup 1 --variants 0 --pmc 0 --no-cpu-baseline --no-cpu-ops --settle-ms 100 --miopen-find 1"
python bench.py --workload yolov6s_int4_lsq --steps 8 --batch 8 --repeats 3 $C 2>$O/cfg5.err | tail -1 > $O/cfg5.json
python -c "import json; j=json.load(open('$O/cfg5.json')); print(j['value'], j['values'], j['config']['step'], j['roofline']['frac'], j['roofline']['avg_launch_us'])"
cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/micro_trace -o micro -- python $R/tools/microbench.py --tensors B,Bx32 --only lsq_bwd,fq_linear > $O/micro.txt 2>&1
cd $R
python tools/kernel_times.py $(find $O/micro_trace -name "*kernel_trace.csv" | head -1) > $O/kernel_times_micro.txt 2>&1
rm -rf $O/micro_trace
grep -v amdgpu $O/micro.txt; cat $O/kernel_times_micro.txt
