cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in "" 1 "" 1; do
  if [ -n "$v" ]; then export SERT_SEG_NO_FUSED_UPPER=1; else unset SERT_SEG_NO_FUSED_UPPER; fi
  python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('no_fused_upper=[$v] ms/step %.4f segsum %.2f loss %.6f' % (d['ms_per_step'], k['word_grad_segsum']['us'], d['last_loss']))"
done
unset SERT_SEG_NO_FUSED_UPPER
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | tail -4
