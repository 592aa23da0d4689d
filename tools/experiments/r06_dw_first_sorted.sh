# Round 6, item 15: dW / db FIRST on the side stream for the SORTED entity chain too (V_e > 2048, small R_e: the reference's product-search settings),
# now that the key sort runs beside the forward and the chain behind the fork is short.  SERT_DW_FIRST=2 forces it (variants library).
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06v
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-12s %-8s ms/step %.4f loss %.6f in-step us: sort %.1f reduce %.1f dW %.1f tree %.1f update %.1f tail %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('entity_sort', 0), k.get('entity_grad_reduce', 0), k.get('gemm_dW', 0), k.get('word_grad_segsum', 0), k.get('optimizer_word_table', 0), k.get('finalize', 0)))"
}
for rep in 1 2 3; do for v in main side; do
  TAGV=$v; unset SERT_DW_FIRST
  [ $v = side ] && export SERT_DW_FIRST=2
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run ps_1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
  run ps_16384 --batch 16384 --entities 32768 --dim 300 --entity-dim 128
  run d128_16k --batch 16384 --entities 32768 --dim 128
  run d128_64k --batch 65536 --entities 32768 --dim 128
done; done 2>&1 | tee $R/gpurun_out/r06v/dw_first_sorted.txt
