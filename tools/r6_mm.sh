#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
for v in ppq_amd/libppq_hip.so variants/lib_mmnostream.so; do
  echo "== $v"
  PPQHIP_LIBRARY=$R/$v python tools/north_star.py --mode trace --sizes 2,4,8,16,32 --report /tmp/mm.txt > /dev/null 2>&1
  grep -E "minmax_t |floor_read " /tmp/mm.txt | cut -c1-130
done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_calibration.py -q -m gpu -x -k "minmax or slots" 2>&1 | tail -3
