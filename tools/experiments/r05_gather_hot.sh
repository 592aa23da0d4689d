# the forward's gather with the batch's hot rows (the dense heavy words, <= 16) staged in LDS (product) against the plain kernel
# (the default; SERT_GATHER_HOT=1 switches the LDS form on); A/B x 3 on one box
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-6s ms/step %.4f  gather %.1f us' % ('$name', '$TAGV', r['ms_per_step'], k.get('gather', 0)))"
}
for rep in 1 2 3; do for v in hot plain; do
  TAGV=$v; unset SERT_GATHER_HOT
  [ $v = hot ] && export SERT_GATHER_HOT=1
  run c2 --batch 65536
  run c2_8192 --batch 8192
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  STEPS=60 run c4 --vocab 500000 --entities 100000 --dim 300
done; done
