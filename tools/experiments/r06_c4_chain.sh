# Round 6, item 13: C4 -- the key sort beside the forward (forced: SERT_EARLY_SORT=1) with the rest of the sorted chain forked behind the word
# gradient's tree (SERT_CHAIN_BEHIND_TREE=1) instead of behind the loss kernel; variants library
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06q
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-60} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-14s ms/step %.4f loss %.6f in-step us: sort %.1f reduce %.1f fixup %.1f dW %.1f tree %.1f update %.1f tail %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('entity_sort', 0), k.get('entity_grad_reduce', 0), k.get('entity_grad_fixup', 0), k.get('gemm_dW', 0), k.get('word_grad_segsum', 0), k.get('optimizer_word_table', 0), k.get('finalize', 0)))"
}
for rep in 1 2 3; do for v in product early early_behind_tree; do
  TAGV=$v; unset SERT_EARLY_SORT SERT_CHAIN_BEHIND_TREE
  [ $v = early ] && export SERT_EARLY_SORT=1
  [ $v = early_behind_tree ] && export SERT_EARLY_SORT=1 SERT_CHAIN_BEHIND_TREE=1
  run c4 --vocab 500000 --entities 100000 --dim 300
  run c4_32k --batch 32768 --vocab 500000 --entities 100000 --dim 300
done; done 2>&1 | tee $R/gpurun_out/r06q/c4_chain.txt
