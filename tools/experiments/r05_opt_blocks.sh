# grid of the streaming optimiser launches: 2048 workgroups (product) against 1792 = 7 per CU (what fits at 72 registers) and 3584
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s blocks=%-5s ms/step %.4f  word-table update %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2 3; do for v in 2048 1792 3584; do
  TAGV=$v
  if [ $v = 2048 ]; then unset SERT_LIB; else export SERT_LIB=$R/sert_amd/variants/libsert_ob$v.so; fi
  run c2 --batch 65536
  run c2_8192 --batch 8192
done; done
