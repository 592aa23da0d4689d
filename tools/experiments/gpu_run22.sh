cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in 1 0 1 0; do
  SERT_SEG_UNITS=$v python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('units=$v ms/step %.4f' % d['ms_per_step'], {a:b['us'] for a,b in k.items()})"
done
for v in 1 0; do
SERT_SEG_UNITS=$v python tools/bench_c4.py --kinds vectorspace --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())['vectorspace']
print('C4 units=$v ms/step %.4f' % d['ms_per_step'], d['kernels_us'])"
done
