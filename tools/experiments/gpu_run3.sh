set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "vectorspace" > gpurun_out/r02c/pytest_vs.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02c/pytest_vs.log
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for sortv in 1 0; do
  for rep in 1 2; do
  SERT_EGRAD_SORT=$sortv python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('egrad_sort=$sortv ms/step %.4f Mpairs/s %.1f | sort %.1f egrad %.1f fix %.1f' % (d['ms_per_step'], d['value']/1e6, k.get('entity_sort',{}).get('us',0), k['entity_grad_reduce']['us'], k['entity_grad_fixup']['us']))"
  done
done 2>&1 | grep -v "^+" | tee gpurun_out/r02c/variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02c/pytest.log
