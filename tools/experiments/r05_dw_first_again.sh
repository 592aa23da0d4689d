# dW / db first on the side stream at batch 65536 too (SERT_DW_FIRST=2, variants library), now that the tree fetches fewer rows
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --num-batches 8 "$@" --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-10s dw_first=%-2s ms/step %.4f' % ('$name', '${SERT_DW_FIRST:--}', r['ms_per_step']))"
}
for rep in 1 2 3; do for f in - 2; do
  if [ $f = - ]; then unset SERT_DW_FIRST; else export SERT_DW_FIRST=$f; fi
  run c2 --batch 65536
  run c2_48k --batch 49152
done; done
