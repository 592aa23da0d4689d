# small batches: the deferred R_e update costs the next loss kernel a cross-queue wait; in front of the tail (SERT_RE_DEFER=0) the
# tail waits for the side chain instead.  A/B/A/B per batch size, C2 dims.
R=$GRAFT_REPO_ROOT
for b in 4096 8192 16384 32768; do for rep in 1 2; do for dfr in 1 0; do
  SERT_RE_DEFER=$dfr python $R/bench.py --num-batches 8 --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('batch $b SERT_RE_DEFER=$dfr ms/step %.4f' % r['ms_per_step'])"
done; done; done
