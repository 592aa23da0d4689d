cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl23
mkdir -p $OUT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
cd /tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt4 -o kt -- python $GRAFT_REPO_ROOT/tools/bench_c4.py --kinds vectorspace --steps 10 > $OUT/run_c4.json 2> $OUT/kt4.err
DB=$(find $OUT/kt4 -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $DB vs_gather_mean 14 > $OUT/timeline_c4.txt 2>&1
rm -rf $OUT/kt4
cat $OUT/timeline_c4.txt
