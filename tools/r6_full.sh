#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite, smoke(), the default bench, the north-star curve
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r6full}
rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -q -m gpu -x) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
(time python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
(time timeout 600 python tools/north_star.py --mode both --report $O/frac_vs_size.txt --json $O/north_star.json) > $O/north_star_scalars.json 2> $O/north_star.err
cut -c1-150 $O/frac_vs_size.txt | grep -vE "floor_copy|fq_linear_t|asym"
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print({k:j[k] for k in ('value','ms_per_step','spread_pct')}); print(j['roofline'])
print({k:v for k,v in c.items() if k.startswith('B_') and ('median' in k or 'status' in k)})
print({k:v for k,v in c.items() if 'seam' in k or 'cfg' in k or 'batch1' in k or 'percentile' in k or 'kl4096' in k})"
