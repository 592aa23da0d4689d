# Round 6, item 7: dW / db FIRST on the side stream (beside dh and the tree) at batch 65536 and above, now that the entity keys' partition
# runs beside the forward (item 1) -- round 5 measured a wash at 65536 (r05 item 20).  SERT_DW_FIRST=2 forces it (variants library).
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06i
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-16s ms/step %.4f  in-step us: update %.1f tail %.1f tree %.1f dW %.1f egrad %.1f bucket %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('entity_grad_reduce', 0), k.get('entity_sort', 0)))"
}
for rep in 1 2 3; do for v in main_early side_early main_behind side_behind; do
  TAGV=$v; unset SERT_DW_FIRST SERT_EARLY_BUCKET
  case $v in side_*) export SERT_DW_FIRST=2;; esac
  case $v in *_behind) export SERT_EARLY_BUCKET=0;; *_early) export SERT_EARLY_BUCKET=1;; esac
  run c2 --batch 65536
  run c2_131k --batch 131072
  run c2_49152 --batch 49152
done; done 2>&1 | tee $R/gpurun_out/r06i/dw_first_again.txt
