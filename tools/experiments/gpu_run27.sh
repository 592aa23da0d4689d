cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in 0 2 1 0 2; do
  SERT_STRIP_GEMM=$v python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('strip=$v ms/step %.4f loss %.6f' % (d['ms_per_step'], d['last_loss']), {a:b['us'] for a,b in k.items() if a.startswith('gemm')})"
done
