set -x
python tools/hist_variants.py > gpurun_out/r2_variants2.log 2>&1; grep insitu gpurun_out/r2_variants2.log
cp gpurun_out/hist_variants.json gpurun_out/hist_variants2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_vtrace -o v -- python $GRAFT_REPO_ROOT/tools/hist_variants.py b512w2u2.so > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r2_vtrace -name "*.db" -delete; ls -R $GRAFT_REPO_ROOT/gpurun_out/r2_vtrace | head
