# level 0 of the word-gradient tree sorted by item length + a one-entry item's row number in its descriptor (product) against the
# items in word order (SERT_SEG_NO_SORT=1); variants library, A/B x 3 on one box
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-10s ms/step %.4f  word_grad_segsum %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('word_grad_segsum', 0)))"
}
for rep in 1 2 3; do for v in sorted word_order; do
  TAGV=$v; unset SERT_SEG_NO_SORT
  [ $v = word_order ] && export SERT_SEG_NO_SORT=1
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done
