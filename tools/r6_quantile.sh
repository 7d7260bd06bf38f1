#!/bin/bash
# Runs ON THE GPU BOX: the two-launch hinted quantile -- its tests, the soak under uneven load, and its timings on B .. Bx32.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6q}
rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "quantile" --durations=8) > $O/pytest_quantile.txt 2>&1
tail -14 $O/pytest_quantile.txt
(time timeout 600 python tools/quantile_soak.py 60 1) > $O/soak.txt 2>&1; tail -3 $O/soak.txt
(time timeout 600 python tools/quantile_soak.py 30 2 big) > $O/soak_big.txt 2>&1; tail -3 $O/soak_big.txt
(time timeout 600 python tools/north_star.py --mode both --sizes ${2:-1,2,8,32} --report $O/frac_vs_size.txt --json $O/north_star.json) > $O/north_star_scalars.json 2> $O/north_star.err
grep -E "status|quantile|floor_read |floor_empty" $O/frac_vs_size.txt | cut -c1-200
