# dense_update_skip with the next row's pieces requested ahead of the current row's arithmetic (SERT_SKIP_PF=1, the product)
# against the same kernel without (variants library built with -DSERT_SKIP_PF=0); A/B/A/B on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-4s ms/step %.4f  word-table update %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2; do for v in pf1 pf0; do
  TAGV=$v
  if [ $v = pf0 ]; then export SERT_LIB=$R/sert_amd/variants/libsert_skip_pf0.so; else unset SERT_LIB; fi
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run w3c --model loglinear --batch 1024 --window 8 --entities 715 --dim 300
  run c4 --vocab 500000 --entities 100000 --dim 300
done; done
