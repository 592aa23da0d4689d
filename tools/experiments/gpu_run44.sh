cd /tmp
export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
OUT=$GRAFT_REPO_ROOT/gpurun_out/t44
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 $NOX > /dev/null 2> $OUT/err
DB=$(find $OUT/kt -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB | grep -i "segsum" | cut -c1-200
rm -rf $OUT/kt
