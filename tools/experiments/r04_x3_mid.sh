# mid-size shapes that gemm_x3.h's 128 x 128-tile form now takes (x3_mid_size) against the fp32 MFMA kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in x3 fp32; do
  if [ $v = fp32 ]; then export SERT_GEMM_FP32=1; else unset SERT_GEMM_FP32; fi
  echo "== $v"
  python - <<'PY'
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
for name, kw in [('NN 16384x300x300 tanh', dict(M=16384, N=300, K=300, epi=2)), ('NT 16384x300x300', dict(M=16384, N=300, K=300, tb=1)),
                 ('NN 8192x300x300 tanh', dict(M=8192, N=300, K=300, epi=2)), ('NT 8192x300x300', dict(M=8192, N=300, K=300, tb=1)),
                 ('NN 4096x300x300 tanh', dict(M=4096, N=300, K=300, epi=2)), ('NT 4096x300x300', dict(M=4096, N=300, K=300, tb=1)),
                 ('NN 8192x128x300 tanh', dict(M=8192, N=128, K=300, epi=2)), ('NN 2033x715x300 bias', dict(M=2033, N=715, K=300, epi=1)),
                 ('NN 5000x1000x256', dict(M=5000, N=1000, K=256)), ('NT 12288x256x256', dict(M=12288, N=256, K=256, tb=1)),
                 ('NN 2300x100000x300 bias', dict(M=2300, N=100000, K=300, epi=1, iters=5))]:
    us = C.bench_gemm(**kw)
    print('%-30s %8.1f us' % (name, us))
PY
done
