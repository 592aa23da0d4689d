# ms/step of the configurations round 5 watches (headline C2 + the small-batch ones); extra env is passed through
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-10s ms/step %.4f  deferred %.4f   %s' % ('$name', r['ms_per_step'], (r.get('deferred_loss_readback') or {}).get('ms_per_step', 0), ' '.join('%s %.1f' % (a[:14], b) for a, b in list(k.items())[:6])))"
}
for rep in 1 2; do
run c2 --batch 65536
run c2_8192 --batch 8192
run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
run w3c --model loglinear --batch 1024 --window 8 --entities 715 --dim 300
done
