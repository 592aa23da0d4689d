# round-5 verdict item 7: the batch-independent half of the C2 word-table update (the rows the batch does not touch: g = lambda/B p)
# on a stream of its own beside the forward (SERT_ADAM_SPLIT=1, variants build), A/B/A/B on one box
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
for rep in 1 2; do for s in 0 1; do
  SERT_ADAM_SPLIT=$s python $R/bench.py --num-batches 8 --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('C2 SERT_ADAM_SPLIT=$s ms/step %.4f  loss %.6f' % (r['ms_per_step'], r['last_loss']))"
done; done
