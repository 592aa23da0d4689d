# Round 6, item 10: the split count of dW now that the GEMM runs on the side stream (off the critical path) and its combine in the tail on the
# main stream (on it): fewer k ranges = fewer partial slabs for the tail to read.  SERT_DW_SPLITS (variants library).
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06n
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-10s ms/step %.4f  in-step us: update %.1f tail %.1f tree %.1f dW %.1f egrad %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('entity_grad_reduce', 0)))"
}
for rep in 1 2; do for v in 256 128 64 32; do
  TAGV=splits_$v; export SERT_DW_SPLITS=$v
  run c2 --batch 65536
  run c2_32768 --batch 32768
  run c2_8192 --batch 8192
done; done 2>&1 | tee $R/gpurun_out/r06n/dw_splits.txt
