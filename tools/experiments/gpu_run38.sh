cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in nce dh nce dh; do
  SERT_FORK_AT=$v python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('fork_at=$v ms/step %.4f loss %.6f' % (d['ms_per_step'], d['last_loss']))"
done
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl38
mkdir -p $OUT
cd /tmp
SERT_FORK_AT=nce timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 $NOX > $OUT/run.json 2> $OUT/kt.err
DB=$(find $OUT/kt -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $DB vs_gather_mean 3 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
cat $OUT/timeline.txt
