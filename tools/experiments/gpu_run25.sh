cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for B in 1024 4096 16384; do
for v in 2 1; do
  SERT_STREAMS=$v python bench.py --batch $B --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('B=$B streams=$v ms/step %.4f  sum_alone %.1f us' % (d['ms_per_step'], sum(b['us'] for b in k.values())), {a:b['us'] for a,b in k.items()})"
done
done
for v in 2 1; do
SERT_STREAMS=$v python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('product-search B=4096 streams=$v ms/step %.4f  sum_alone %.1f us' % (d['ms_per_step'], sum(b['us'] for b in k.values())), {a:b['us'] for a,b in k.items()})"
done
