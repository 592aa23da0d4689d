# Round 6, item 14: the split-K combine of dW / db on the side stream behind the GEMM (the tail then reads 66 kB of sums instead of the slabs)
# against inside the tail (SERT_COMBINE_SIDE=0, variants library)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06t
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-8s ms/step %.4f loss %.6f in-step us: update %.1f tail %.1f combine %.1f tree %.1f dW %.1f egrad %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('splitk_combine', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('entity_grad_reduce', 0)))"
}
for rep in 1 2 3; do for v in side tail; do
  TAGV=$v; unset SERT_COMBINE_SIDE
  [ $v = tail ] && export SERT_COMBINE_SIDE=0
  run c2 --batch 65536
  run c2_32768 --batch 32768
  run c2_16384 --batch 16384
  run c2_8192 --batch 8192
  run c2_2048 --batch 2048
done; done 2>&1 | tee $R/gpurun_out/r06t/combine_side.txt
