#!/bin/bash
# Runs ON THE GPU BOX: parity of the touched kernels + interleaved A/B (HEAD vs round 4) on B, Bx8, Bx32 and A.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_floor3}
rm -rf $O; mkdir -p $O
run() {  # tag, args...
  tag=$1; shift
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$tag -o floor -- python $R/tools/floor_table.py --tag $tag "$@" > $O/l2l_$tag.txt 2>&1
  cd $R
  csv=$(find $O/trace_$tag -name "*kernel_trace.csv" | head -1)
  python tools/floor_report.py "$csv" gpurun_out/floor_manifest_$tag.json > $O/floor_$tag.txt 2>&1
  cp gpurun_out/floor_manifest_$tag.json $O/
  rm -rf $O/trace_$tag
}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py "tests/test_gpu_calibration.py::test_per_channel_activation_observers_under_hip_graph_capture" "tests/test_gpu_calibration.py::test_graph_and_async_modes_equal_eager" tests/test_gpu_finetune.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
run b --rounds 3 --libs r04=variants/lib_r04.so
run bx8 --product-only --rounds 2 --iters 60 --shape 8,512,56,56 --libs r04=variants/lib_r04.so
run bx32 --product-only --rounds 2 --iters 30 --shape 32,512,56,56 --libs r04=variants/lib_r04.so
run a --product-only --rounds 2 --shape 1,3,224,224 --libs r04=variants/lib_r04.so
for t in b bx8 bx32 a; do echo "== $t"; grep -v "^#" $O/floor_$t.txt | cut -c3-75,92-; done
