#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of the library builds on B (one process, same clocks), 3 rounds.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_floor2}
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
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py -x -q -m gpu > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
LIBS="r04=variants/lib_r04.so,dev=variants/lib_dev.so,fqU1=variants/lib_fqU1.so,fqnorcp=variants/lib_fqnorcp.so,fqU1norcp=variants/lib_fqU1norcp.so"
run ab --product-only --rounds 3 --libs $LIBS --hist-wg 48,64,80,98,112,128,160,196 --rows-wg 128,196,256,392,512
HIP_FORCE_DEV_KERNARG=1 run devkernarg --rounds 2 --libs dev=variants/lib_dev.so --hist-wg 98,128 --rows-wg 256,392
HIP_FORCE_DEV_KERNARG=0 run hostkernarg --rounds 2 --libs dev=variants/lib_dev.so --hist-wg 98,128 --rows-wg 256,392
for t in ab devkernarg hostkernarg; do echo "== $t"; grep -v "^#" $O/floor_$t.txt | cut -c1-75,92-; done
