cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for B in 8192 16384 32768; do
for v in "" 1; do
  if [ -n "$v" ]; then export SERT_SEG_NO_FUSED_UPPER=1; else unset SERT_SEG_NO_FUSED_UPPER; fi
  python bench.py --batch $B --steps 300 --warmup 30 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('B=$B no_fused_upper=[$v] ms/step %.4f segsum %.2f' % (d['ms_per_step'], k['word_grad_segsum']['us']))"
done
done
