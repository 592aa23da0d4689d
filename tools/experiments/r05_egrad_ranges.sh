# entity gradient of few pairs over a mid-size table: the one-launch range kernel against sort + chunked reduce + fix-up (SERT_EGRAD_SORT=1)
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 300 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-14s ranges=%s ms/step %.4f  sort %.1f reduce %.1f fixup %.1f  loss %.6f' % ('$name', '${SERT_EGRAD_RANGES:-0}', r['ms_per_step'], k.get('entity_sort',0), k.get('entity_grad_reduce',0), k.get('entity_grad_fixup',0), r['last_loss']))"
}
for rep in 1 2; do for srt in 0 1; do
if [ $srt = 1 ]; then unset SERT_EGRAD_RANGES; else export SERT_EGRAD_RANGES=1; fi
run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
run ps_b1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
run c4tables_b4096 --batch 4096 --vocab 500000 --entities 100000 --dim 300
done; done
