# gemm_direct.h against gemm.h, shape by shape (variants build: SERT_GEMM_DIRECT_MIN_K=1000000 switches it off)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SERT_LIB=$R/sert_amd/variants/libsert_variants.so
for v in direct tiled; do
  if [ $v = tiled ]; then export SERT_GEMM_DIRECT_MIN_K=1000000; else unset SERT_GEMM_DIRECT_MIN_K; fi
  echo "== $v"
  python - <<'PY'
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
for name, kw in [('fs logits NT 65536x1000x128 (K<256: tiled)', dict(M=65536, N=1000, K=128, tb=1)), ('fs dp NN 65536x128x1000', dict(M=65536, N=128, K=1000)),
                 ('ll fwd NN 44467x1000x128 (tiled)', dict(M=44467, N=1000, K=128, epi=1)), ('ll dG NT 44467x128x1000', dict(M=44467, N=128, K=1000, tb=1)),
                 ('c4 proj NN 65536x300x300 tanh', dict(M=65536, N=300, K=300, epi=2)), ('c4 dh NT 65536x300x300', dict(M=65536, N=300, K=300, tb=1)),
                 ('4096^3 NN', dict(M=4096, N=4096, K=4096, iters=5)), ('4096^3 NT', dict(M=4096, N=4096, K=4096, tb=1, iters=5)),
                 ('ll (81920) dG NT 81920x128x1000', dict(M=81920, N=128, K=1000, tb=1)), ('query-like NT 10000x100000x128 (tiled)', dict(M=10000, N=100000, K=128, tb=1, iters=3))]:
    us = C.bench_gemm(**kw)
    print('%-46s %8.1f us %7.1f TF' % (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
PY
done
