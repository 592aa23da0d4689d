# Round 6, item 9: timing knock-out (WRONG loss) -- the tail split in two: dW combine + W, b Adam right behind dW on the side stream (the next
# projection waits for its event), the loss as a one-workgroup launch behind the word table's update (in the real thing: that kernel's last workgroup)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06l
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-10s ms/step %.4f  in-step us: update %.1f tail %.1f tree %.1f dW %.1f egrad %.1f' % ('$name', '$TAGV', r['ms_per_step'], k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('word_grad_segsum', 0), k.get('gemm_dW', 0), k.get('entity_grad_reduce', 0)))"
}
for rep in 1 2 3; do for v in product tail_early; do
  TAGV=$v; unset SERT_KO_TAIL_EARLY
  [ $v = tail_early ] && export SERT_KO_TAIL_EARLY=1
  run c2 --batch 65536
  run c2_32768 --batch 32768
  run c2_16384 --batch 16384
  run c2_8192 --batch 8192
  run c2_2048 --batch 2048
done; done 2>&1 | tee $R/gpurun_out/r06l/ko_tail_early.txt
