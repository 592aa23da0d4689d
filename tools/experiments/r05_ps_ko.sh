# what binds the product-search step: knock the entity chain out (SERT_KO_EGRAD: wrong results, timing only), fork at the loss kernel, both
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-8s %-40s ms/step %.4f' % ('$name', '$TAG', r['ms_per_step']))"
}
PS="--batch 4096 --entities 32768 --dim 300 --entity-dim 128"
for rep in 1 2; do
  TAG=base run ps $PS
  TAG=ko_egrad SERT_KO_EGRAD=1 run ps $PS
  TAG=fork_nce SERT_FORK_AT=nce run ps $PS
  TAG=fork_nce+ko_egrad SERT_FORK_AT=nce SERT_KO_EGRAD=1 run ps $PS
  TAG=no_defer SERT_RE_DEFER=0 run ps $PS
  TAG=lazy_never SERT_LAZY_MAX=0 run ps $PS
done
