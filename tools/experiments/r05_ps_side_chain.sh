# product-search settings: the loop loss(t) -> dh -> fork -> entity chain -> R_e update -> loss(t+1) is the step's period now.
# egrad_ranges (one launch instead of eight, SERT_EGRAD_RANGES=1) and the fork behind the loss kernel (SERT_FORK_AT=nce), variants library
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-300} --warmup 24 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-14s ms/step %.4f' % ('$name', '$TAGV', r['ms_per_step']))"
}
PS="--batch 4096 --entities 32768 --dim 300 --entity-dim 128"
for rep in 1 2 3; do for v in default ranges fork_nce ranges+fork; do
  TAGV=$v; unset SERT_EGRAD_RANGES SERT_FORK_AT
  case $v in ranges) export SERT_EGRAD_RANGES=1;; fork_nce) export SERT_FORK_AT=nce;; ranges+fork) export SERT_EGRAD_RANGES=1 SERT_FORK_AT=nce;; esac
  run ps $PS
  run ps1024 --batch 1024 --entities 32768 --dim 300 --entity-dim 128
done; done
