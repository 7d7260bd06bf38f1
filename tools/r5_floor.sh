#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the floor table of round 5 + the A/B variants of the single-tensor launches on B.
#   gpurun --timeout 1500 -- 'bash tools/r5_floor.sh [outdir-tag]'
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_floor}
rm -rf $O; mkdir -p $O
run() {  # tag, extra args...; env PPQHIP_LIBRARY is inherited
  tag=$1; shift
  cd /tmp
  timeout 400 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$tag -o floor -- python $R/tools/floor_table.py --tag $tag "$@" > $O/l2l_$tag.txt 2>&1
  cd $R
  csv=$(find $O/trace_$tag -name "*kernel_trace.csv" | head -1)
  python tools/floor_report.py "$csv" gpurun_out/floor_manifest_$tag.json > $O/floor_$tag.txt 2>&1
  cp gpurun_out/floor_manifest_$tag.json $O/
  rm -rf $O/trace_$tag
}
# 0. parity of the changed kernels first
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_reference.py -x -q -m gpu > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
# 1. floor kernels + the library at HEAD
run head
# 2. the library as of round 4, and the variants
for v in r04 fqU1 fqU4 fqnorcp fqU1norcp; do
  [ -f variants/lib_$v.so ] && PPQHIP_LIBRARY=$R/variants/lib_$v.so run $v --product-only
done
[ -f variants/lib_dev.so ] && PPQHIP_LIBRARY=$R/variants/lib_dev.so run dev --product-only --hist-wg 32,64,98,128,160,196,256
# 3. B x 32 must not regress: launch-to-launch over rotating buffers (microbench), HEAD vs r04
timeout 300 python tools/microbench.py --tensors Bx32 --only fq_linear,hist_sym_t,minmax_t > $O/micro_bx32_head.txt 2>&1
PPQHIP_LIBRARY=$R/variants/lib_r04.so timeout 300 python tools/microbench.py --tensors Bx32 --only fq_linear,hist_sym_t,minmax_t > $O/micro_bx32_r04.txt 2>&1
cat $O/floor_head.txt
for v in r04 fqU1 fqU4 fqnorcp fqU1norcp dev; do echo "== $v"; grep -v "^#" $O/floor_$v.txt | cut -c1-60,92-; done
grep -v amdgpu $O/micro_bx32_head.txt; grep -v amdgpu $O/micro_bx32_r04.txt
