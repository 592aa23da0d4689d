# segsum_rows: the row numbers of the next 32 / 64 entries requested before the current ones are walked (product) against the library
# of the commit before (sert_amd/variants/libsert_before.so); A/B x 3 on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-8s ms/step %.4f  tree %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('word_grad_segsum', 0) or k.get('per_word_dz_sums', 0)))"
}
for rep in 1 2 3; do for v in prefetch before; do
  TAGV=$v
  if [ $v = before ]; then export SERT_LIB=$R/sert_amd/variants/libsert_before.so; else unset SERT_LIB; fi
  run c2 --batch 65536
  run c2_8192 --batch 8192
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
  STEPS=60 run ll --model loglinear --batch 65536
done; done
