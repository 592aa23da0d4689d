# where does the fused gather + projection kernel's time go?  knock-outs (variant libraries, -DPJ_KO=k; wrong results)
R=$GRAFT_REPO_ROOT
for lib in product pj_ko1 pj_ko2; do
  if [ $lib = product ]; then L=$R/sert_amd/libsert_hip.so; else L=$R/sert_amd/variants/libsert_$lib.so; fi
  for b in 65536 8192; do
  SERT_LIB=$L python $R/bench.py --num-batches 8 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r.get('kernel_us',{})
print('$lib batch $b: fused kernel %.1f us' % k.get('gather',0))"
  done
done
