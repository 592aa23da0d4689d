cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for B in 1024 4096 16384 65536; do
  python bench.py --batch $B --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('B=$B ms/step %.4f' % (d['ms_per_step']), {a:b['us'] for a,b in k.items() if a.startswith('entity')})"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | tail -8
