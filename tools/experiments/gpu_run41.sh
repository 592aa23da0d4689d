cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for w in 1 2 3; do
  SERT_STRIP_GEMM=2 SERT_STRIP_WGS=$w python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('roles wgs/CU=$w ms/step %.4f loss %.6f' % (d['ms_per_step'], d['last_loss']), {a:b['us'] for a,b in k.items() if a.startswith('gemm')})"
done
