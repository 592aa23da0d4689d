# Round 6, item 17: the step's tail on a queue of its own (beside the next step's gather; the next projection waits for it) against on the
# main stream (SERT_TAIL_QUEUE=0, variants library)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
mkdir -p $R/gpurun_out/r06y
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us_instep',{})
print('%-9s %-6s ms/step %.4f loss %.6f in-step us: gather %.1f fwd %.1f update %.1f tail %.1f tree %.1f' % ('$name', '$TAGV', r['ms_per_step'], r['last_loss'], k.get('gather', 0), k.get('gemm_fwd', 0), k.get('optimizer_word_table', 0), k.get('finalize', 0), k.get('word_grad_segsum', 0)))"
}
for rep in 1 2 3; do for v in queue main; do
  TAGV=$v; unset SERT_TAIL_QUEUE
  [ $v = main ] && export SERT_TAIL_QUEUE=0
  run c2 --batch 65536
  run c2_32768 --batch 32768
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done 2>&1 | tee $R/gpurun_out/r06y/tail_queue.txt
