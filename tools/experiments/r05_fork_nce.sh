# where the entity chain forks off the main stream: behind the dh GEMM (default) or behind the loss kernel (SERT_FORK_AT=nce).
# A/B/A/B on one box with the -DSERT_VARIANTS library; product-search settings, C2 dims at 8192 / 65536 rows.
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-10s fork=%-4s ms/step %.4f' % ('$name', '${SERT_FORK_AT:-dh}', r['ms_per_step']))"
}
for rep in 1 2; do for f in dh nce; do
  export SERT_FORK_AT=$f
  run ps --batch 4096 --entities 32768 --dim 300 --entity-dim 128
  run ps1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
  run c2_8192 --batch 8192
  run c2 --batch 65536
done; done
