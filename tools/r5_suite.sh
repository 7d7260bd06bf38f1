#!/bin/bash
# Runs ON THE GPU BOX: the full GPU suite, then an interleaved A/B of fq tile shapes on B, then the driver's bench command.
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5_suite}
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests -q -m gpu -x) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
LIBS=""
for v in r04 fqS2 fqS4; do [ -f variants/lib_$v.so ] && LIBS="$LIBS,$v=variants/lib_$v.so"; done
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace_ab -o floor -- python $R/tools/floor_table.py --tag ab --rounds 3 --libs ${LIBS#,} --cases fq_linear > $O/l2l_ab.txt 2>&1
cd $R
python tools/floor_report.py "$(find $O/trace_ab -name '*kernel_trace.csv' | head -1)" gpurun_out/floor_manifest_ab.json > $O/floor_ab.txt 2>&1
rm -rf $O/trace_ab
grep -v "^#" $O/floor_ab.txt | grep "lib\[\|floor_copy\|floor_empty grid=256 block=256\|floor_read block=256 U=2 grid=one" | cut -c3-75,92-
(time python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-1500
tail -5 $O/bench.err
