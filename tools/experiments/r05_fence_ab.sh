# cross-stream events without the system-scope fence (hipEventDisableSystemFence), A/B on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('%-10s fence=%-7s ms/step %.4f  deferred %.4f  loss %.6f' % ('$name', '${SERT_EVENT_FENCE:-device}', r['ms_per_step'], (r.get('deferred_loss_readback') or {}).get('ms_per_step', 0), r['last_loss']))"
}
for rep in 1 2; do
for f in device system; do
if [ $f = system ]; then export SERT_EVENT_FENCE=system; else unset SERT_EVENT_FENCE; fi
run c2 --batch 65536
run c2_8192 --batch 8192
run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
done; done
