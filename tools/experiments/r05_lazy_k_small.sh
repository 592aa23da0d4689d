# ... and fewer: a full pass every 2nd / 3rd update (less catch-up and prediction arithmetic, more bytes)
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s K=%-3s ms/step %.4f  word-table update %.1f us' % ('$name', '$K', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2; do for K in 4 3 2; do
  if [ $K = 4 ]; then unset SERT_LIB; else export SERT_LIB=$R/sert_amd/variants/libsert_k$K.so; fi
  run c2 --batch 65536
  run c2_32k --batch 32768
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done
