cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in 1 0 1 0; do export SERT_SEG_UNITS=$v;
  python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('seg_units=$v ms/step %.4f' % d['ms_per_step'])"
done
unset SERT_SEG_UNITS
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl21
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 $NOX > $OUT/run.json 2> $OUT/kt.err
DB=$(find $OUT/kt -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $DB vs_gather_mean 3 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
cat $OUT/timeline.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | tail -8
