# loglinear per-word dZ sums: the dense heavy words inside the tree's launches (segsum_rows_plus_ll) against the two launches behind
# the tree (SERT_HEAVY_NO_FUSE=1); variants library, A/B x 3 on one box
R=$GRAFT_REPO_ROOT
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
run() { name=$1; shift
  python $R/bench.py --model loglinear --num-batches 8 "$@" --steps ${STEPS:-60} --warmup 6 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-small-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('%-8s %-12s ms/step %.4f  %s' % ('$name', '$TAGV', r['ms_per_step'], ' '.join('%s %.1f' % (a[:18], b) for a, b in list(k.items())[:5])))"
}
for rep in 1 2 3; do for v in fused two_launches; do
  TAGV=$v; unset SERT_HEAVY_NO_FUSE
  [ $v = two_launches ] && export SERT_HEAVY_NO_FUSE=1
  run ll_c2 --batch 65536
  run ll_16k --batch 16384
done; done
