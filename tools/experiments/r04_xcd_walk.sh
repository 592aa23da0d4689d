# XCD-aware item order of the persistent GEMM kernels, A/B on a variants build (SERT_GEMM_NO_XCD_WALK=1 = the old order)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
for v in walk nowalk; do
  if [ $v = nowalk ]; then export SERT_GEMM_NO_XCD_WALK=1; else unset SERT_GEMM_NO_XCD_WALK; fi
  echo "== $v"
  python tools/bench_gemm.py 2>&1
  python - <<'PY'
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
for name, kw in [('c4 proj NN 65536x300x300 tanh', dict(M=65536, N=300, K=300, epi=2)), ('c4 dh NT', dict(M=65536, N=300, K=300, tb=1)),
                 ('c4 dW TN split 114', dict(M=300, N=300, K=65536, ta=1, splits=114)),
                 ('fs logits NT 65536x1000x128', dict(M=65536, N=1000, K=128, tb=1)), ('fs dp NN 65536x128x1000', dict(M=65536, N=128, K=1000)),
                 ('fs dRe TN 1000x128x65536 split 128', dict(M=1000, N=128, K=65536, ta=1, splits=128)),
                 ('ll fwd 44467x1000x128', dict(M=44467, N=1000, K=128, epi=1)), ('ll dW TN 128x1000x44467 split 128', dict(M=128, N=1000, K=44467, ta=1, splits=128))]:
    us = C.bench_gemm(**kw)
    print('%-40s %8.1f us %7.1f TF' % (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
PY
  python tools/bench_c4.py --kinds vectorspace --steps 10 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin)['vectorspace']; print('C4 step %.3f ms' % r['ms_per_step'], r['kernels_us'])"
  python bench.py --model loglinear --steps 20 --warmup 3 --no-cpu-baseline --no-loglinear-extra --no-query-extra --no-c4-extra --no-seed-extra --no-live-pmc 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('LL step %.4f ms' % r['ms_per_step'], {k:v['us'] for k,v in r['kernels'].items()})"
done
