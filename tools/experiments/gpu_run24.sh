cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" 1 "" 1; do
  if [ -n "$v" ]; then export SERT_EGRAD_GROUP_SUM=1; else unset SERT_EGRAD_GROUP_SUM; fi
  python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('separate_group_sum=[$v] ms/step %.4f' % d['ms_per_step'], {a:b['us'] for a,b in k.items() if a.startswith('entity') or a.startswith('optimizer_o')})"
done
unset SERT_EGRAD_GROUP_SUM
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | tail -8
