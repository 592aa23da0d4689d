# dense_update_skip at d_w = 300 (75 float4 per row): 32 lanes x 3 columns per row (SERT_SKIP_32X3=1) against 64 lanes x 2; variants
# library, A/B x 3 on one box
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-6s ms/step %.4f  word-table update %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0)))"
}
for rep in 1 2 3; do for v in 64x2 32x3; do
  TAGV=$v; unset SERT_SKIP_32X3
  [ $v = 32x3 ] && export SERT_SKIP_32X3=1
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run w3c --model loglinear --batch 1024 --window 8 --entities 715 --dim 300
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done
