cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" TWO "" TWO; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L SERT_STRIP_GEMM=2 python bench.py --steps 100 --warmup 10 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f' % (d['ms_per_step']), {a:b['us'] for a,b in k.items() if a.startswith('gemm')})"
done
