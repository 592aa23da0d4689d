# Round 6, item 16: the deferred dense update of a LARGE entity table (C4: 30 M elements, 0.84 GB) behind the step's tail instead of beside it
# (SERT_RE_BEHIND_TAIL=0 = round 5, variants library)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06x
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-60} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-8s ms/step %.4f loss %.6f in-step us: gather %.1f fwd %.1f sort %.1f reduce %.1f dW %.1f tree %.1f update %.1f other %.1f tail %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('gather', 0), k.get('gemm_fwd', 0), k.get('entity_sort', 0), k.get('entity_grad_reduce', 0), k.get('gemm_dW', 0), k.get('word_grad_segsum', 0), k.get('optimizer_word_table', 0), k.get('optimizer_other', 0), k.get('finalize', 0)))"
}
for rep in 1 2 3; do for v in behind beside; do
  TAGV=$v; unset SERT_RE_BEHIND_TAIL
  [ $v = beside ] && export SERT_RE_BEHIND_TAIL=0
  run c4 --vocab 500000 --entities 100000 --dim 300
  run c4_32k --batch 32768 --vocab 500000 --entities 100000 --dim 300
  run e100k_d128 --batch 65536 --entities 100000 --dim 128
done; done 2>&1 | tee $R/gpurun_out/r06x/re_behind_tail.txt
