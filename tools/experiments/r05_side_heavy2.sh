R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-10s side_heavy=%s ms/step %.4f' % ('$name', '${SERT_SIDE_HEAVY:-1}', r['ms_per_step']))"
}
for rep in 1 2; do for f in 1 2; do
  export SERT_SIDE_HEAVY=$f
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run ps1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
done; done
