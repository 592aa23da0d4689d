# Round 6, item 1: egrad_bucket in front of the fork (beside the forward) against behind it (SERT_NO_EARLY_BUCKET=1, variants library);
# A/B x 3 on one box: ms/step + the in-step kernel times of the update and the entity chain (bench.py's kernel_us_instep)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06c
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-8s %-8s ms/step %.4f  in-step us: update %.1f  egrad_acc %.1f  bucket %.1f  tree %.1f  dW %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('entity_grad_reduce', 0), k.get('entity_sort', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0)))"
}
for rep in 1 2 3; do for v in early behind; do
  TAGV=$v; unset SERT_NO_EARLY_BUCKET
  [ $v = behind ] && export SERT_NO_EARLY_BUCKET=1
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run c2_32768 --batch 32768
done; done 2>&1 | tee $R/gpurun_out/r06c/early_bucket.txt
