cd $GRAFT_REPO_ROOT
NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc"
for v in 512 256 384 1024 512; do
  SERT_DW_SPLITS=$v python bench.py --steps 200 --warmup 20 $NOX 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernels']
print('dw_splits=$v ms/step %.4f dW %.2f combine %.2f' % (d['ms_per_step'], k['gemm_dW']['us'], k['splitk_combine']['us']))"
done
