cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in 16 32 64 128 16 32; do
  SERT_EG_GROUPS=$v python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('groups=$v ms/step %.4f' % d['ms_per_step'], {a:b['us'] for a,b in k.items() if a.startswith('entity') or a=='optimizer_other'})"
done
