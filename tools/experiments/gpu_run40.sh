cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" NOPF "" NOPF; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/sert_amd/variants/libsert_$v.so
  SERT_LIB=$L python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('variant=[$v] ms/step %.4f loss %.6f' % (d['ms_per_step'], d['last_loss']), {a:b['us'] for a,b in k.items() if a.startswith('entity')})"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vectorspace" 2>&1 | grep -E "passed|failed" | tail -2
