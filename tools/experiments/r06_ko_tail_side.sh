# Round 6, item 6: timing knock-out (WRONG loss) -- the step's tail (dW combine + W, b Adam + loss) on the side stream behind the entity chain
# instead of on the main stream behind the word table's update: what decoupling its sum of squares from this step's update would buy.
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06h
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-10s ms/step %.4f  in-step us: update %.1f tail %.1f tree %.1f dW %.1f egrad %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('entity_grad_reduce', 0)))"
}
for rep in 1 2 3; do for v in product tail_side; do
  TAGV=$v; unset SERT_KO_TAIL_SIDE SERT_DW_FIRST
  [ $v = tail_side ] && export SERT_KO_TAIL_SIDE=1
  run c2_8192 --batch 8192
  run c2_16384 --batch 16384
  run c2_32768 --batch 32768
  export SERT_DW_FIRST=2
  TAGV=${v}_dwf run c2 --batch 65536
  unset SERT_DW_FIRST
  [ $v = product ] && run c2 --batch 65536
done; done 2>&1 | tee $R/gpurun_out/r06h/ko_tail_side.txt
