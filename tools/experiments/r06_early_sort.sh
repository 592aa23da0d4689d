# Round 6, item 11: the stable sort of the (entity, pair) keys of the sorted entity chain (V_e > 2048) beside the forward, against inside the
# chain behind the fork (SERT_NO_EARLY_SORT=1, variants library): the reference's product-search settings, C4, and two sizes between
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06p
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-8s ms/step %.4f loss %.6f in-step us: sort %.1f reduce %.1f fixup %.1f dW %.1f tree %.1f update %.1f tail %.1f nce %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('entity_sort', 0), k.get('entity_grad_reduce', 0), k.get('entity_grad_fixup', 0), k.get('gemm_dW', 0), k.get('word_grad_segsum', 0), k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('loss', 0)))"
}
for rep in 1 2 3; do for v in early behind; do
  TAGV=$v; unset SERT_NO_EARLY_SORT
  [ $v = behind ] && export SERT_NO_EARLY_SORT=1
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run ps_1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
  run mid --batch 16384 --entities 32768 --dim 128
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done 2>&1 | tee $R/gpurun_out/r06p/early_sort.txt
