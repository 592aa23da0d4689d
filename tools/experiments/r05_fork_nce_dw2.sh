# SERT_FORK_AT=nce_dw again, on top of the fused dense heavy words (variants library; A/B x 3 on one box)
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-10s fork=%-6s ms/step %.4f' % ('$name', '${SERT_FORK_AT:-dh}', r['ms_per_step']))"
}
for rep in 1 2 3; do for f in dh nce_dw; do
  export SERT_FORK_AT=$f
  run c2 --batch 65536
  run c2_128k --batch 131072
done; done
