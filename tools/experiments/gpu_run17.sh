cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" 1 "" 1; do
  if [ -n "$v" ]; then export SERT_NCE_PER_CANDIDATE=1; else unset SERT_NCE_PER_CANDIDATE; fi
  python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('per_candidate=[$v] ms/step %.4f gather %.1f loss %.1f segsum %.1f' % (d['ms_per_step'], k['gather']['us'], k['loss']['us'], k['word_grad_segsum']['us']))"
done
unset SERT_NCE_PER_CANDIDATE
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
