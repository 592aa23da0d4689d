#!/bin/bash
# Profile the bench line's workload on the GPU box:  tools/profile_bench.sh <tag>
#   1. rocprofv3 --kernel-trace --stats           -> profiles/<tag>_vs_c2_kernels.txt
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE    -> profiles/<tag>_vs_c2_pmc.{txt,json}
#      (separate passes, no trace domains besides the kernel trace -- MI355X_MICROARCH.md)
#   3. plain bench.py                             -> profiles/<tag>_bench.json
# Everything is written under gpurun_out/ (merged back by gpurun); copy into profiles/.
set -u
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-query-extra --no-loglinear-extra"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $ROOT/bench.py $ARGS > $OUT/kt.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $ROOT/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python $ROOT/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
cd $ROOT
KT=$(find $OUT/kt -name '*.db' | head -1); FE=$(find $OUT/fetch -name '*.db' | head -1); WR=$(find $OUT/write -name '*.db' | head -1)
python tools/rocpd_summary.py $KT > $OUT/${TAG}_vs_c2_kernels.txt
python tools/rocpd_pmc.py $FE $WR --json $OUT/${TAG}_vs_c2_pmc.json > $OUT/${TAG}_vs_c2_pmc.txt
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
rm -rf $OUT/kt $OUT/fetch $OUT/write
ls -la $OUT
