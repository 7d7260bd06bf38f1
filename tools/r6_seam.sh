#!/bin/bash
# Runs ON THE GPU BOX: attribution of the plain drop-in seam (tools/seam_attribution.py) at batch 32 and batch 1
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6seam}
rm -rf $O; mkdir -p $O
for stack in ${2:-kernels pass}; do
for b in 32 1; do
  steps=20; [ $b = 1 ] && steps=64
  (timeout 600 python tools/seam_attribution.py --batch $b --steps $steps --stack $stack --profile) 2>&1 | grep -vE "amdgpu.ids|Calibration Progress|it/s" | tee -a $O/attribution.txt
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_${stack}_$b -o s -- python $R/tools/seam_attribution.py --batch $b --steps $steps --stack $stack > /dev/null 2>&1
  cd $R
  echo "[$stack] batch $b: kernel trace of the whole process (1 warm + 2 timed passes)" | tee -a $O/attribution.txt
  python tools/seam_kernel_categories.py $(find $O/trace_${stack}_$b -name "*kernel_trace.csv" | head -1) 3 | tee -a $O/attribution.txt
done
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
