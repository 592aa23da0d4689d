# gather + mean-pool + projection in one launch (SERT_PROJ_FUSED=1) at the small batches of C2's dims, final tree; A/B x 3
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 300 --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-8s fused=%s ms/step %.4f' % ('$name', '${SERT_PROJ_FUSED:-0}', r['ms_per_step']))"
}
for rep in 1 2 3; do for f in 0 1; do
  export SERT_PROJ_FUSED=$f
  run c2_2048 --batch 2048
  run c2_4096 --batch 4096
  run c2_8192 --batch 8192
  run c2_16k --batch 16384
done; done
