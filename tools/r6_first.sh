#!/bin/bash
# Runs ON THE GPU BOX: round 6's first look -- the new exhaustive tests with timings, the north-star curve, the whole GPU suite, the default bench.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6a}
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "all_float_patterns" --durations=10) > $O/pytest_exhaustive.txt 2>&1
tail -15 $O/pytest_exhaustive.txt
(time timeout 600 python tools/north_star.py --mode both --report $O/frac_vs_size.txt --json $O/north_star.json) > $O/north_star_scalars.json 2> $O/north_star.err
cat $O/frac_vs_size.txt | cut -c1-150
(time timeout 1800 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_kernels.py -k "not all_float_patterns") > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print({k:j[k] for k in ('value','ms_per_step','spread_pct')}); print(j['roofline'])
print({k:v for k,v in c.items() if k.startswith('B_') or 'crosses' in k or k.startswith('merge_')})"
