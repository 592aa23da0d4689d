# Round 6, item 3: with the entity keys' partition beside the forward, the fork of the (now one-kernel) entity chain behind the LOSS kernel
# (SERT_FORK_AT=nce, variants library) against behind the dh GEMM (default); and the partition forced on / off per batch size.
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06e
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-14s ms/step %.4f  in-step us: update %.1f  egrad_acc %.1f  bucket %.1f  tree %.1f  dW %.1f fwd %.1f nce %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('entity_grad_reduce', 0), k.get('entity_sort', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('gemm_fwd', 0), k.get('loss', 0)))"
}
for rep in 1 2 3; do for v in early_dh early_nce behind_dh; do
  TAGV=$v; unset SERT_FORK_AT SERT_EARLY_BUCKET
  [ $v = early_dh ] && export SERT_EARLY_BUCKET=1
  [ $v = early_nce ] && export SERT_EARLY_BUCKET=1 SERT_FORK_AT=nce
  [ $v = behind_dh ] && export SERT_EARLY_BUCKET=0
  run c2 --batch 65536
  run c2_32768 --batch 32768
  run c2_16384 --batch 16384
  run c2_8192 --batch 8192
done; done 2>&1 | tee $R/gpurun_out/r06e/fork_nce_early.txt
