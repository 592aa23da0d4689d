cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl20
mkdir -p $OUT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 $NOX > $OUT/run.json 2> $OUT/kt.err
DB=$(find $OUT/kt -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $DB vs_gather_mean 3 > $OUT/timeline.txt 2>&1
python tools/rocpd_timeline.py $DB vs_gather_mean 6 >> $OUT/timeline.txt 2>&1
python tools/rocpd_summary.py $DB > $OUT/kernels.txt
rm -rf $OUT/kt
cat $OUT/timeline.txt
