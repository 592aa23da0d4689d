# C2 itself (a batch touches 44 % of the word rows, this or the next batch 69 %) through dense_update_skip instead of adam_l2
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s lazy_max=%-4s ms/step %.4f  word-table update %.1f us' % ('$name', '${SERT_LAZY_MAX:-0.35}', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2 3; do for f in 0.35 1; do
  export SERT_LAZY_MAX=$f
  run c2 --batch 65536
  run c2_32k --batch 32768
  run c2_16k --batch 16384
done; done
