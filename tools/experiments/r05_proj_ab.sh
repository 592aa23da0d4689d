# fused gather + projection (kernels_proj.h), A/B on one box: C2 and C2 dims at 8192 rows
R=$GRAFT_REPO_ROOT
for cfg in "c2 --batch 65536" "c2_8192 --batch 8192"; do
  set -- $cfg; name=$1; shift
  for f in 1 0 1 0; do
    SERT_PROJ_FUSED=$f python $R/bench.py --num-batches 8 "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('$name fused=$f ms/step %.4f  gather %.1f  gemm_fwd %.1f' % (r['ms_per_step'], k.get('gather',0), k.get('gemm_fwd',0)))"
  done
done
