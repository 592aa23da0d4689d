# gemm_x3.h (bf16 pipe, operands split exactly in three) against the fp32 MFMA kernels of gemm.h, shape by shape
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in x3 fp32; do
  if [ $v = fp32 ]; then export SERT_GEMM_FP32=1; else unset SERT_GEMM_FP32; fi
  echo "== $v"
  python - <<'PY'
import sys; sys.path.insert(0, '.')
from sert_amd import _capi as C
for name, kw in [('c2 proj NN 65536x128x128 tanh', dict(M=65536, N=128, K=128, epi=2)), ('c2 dh NT 65536x128x128', dict(M=65536, N=128, K=128, tb=1)),
                 ('c4 proj NN 65536x300x300 tanh', dict(M=65536, N=300, K=300, epi=2)), ('c4 dh NT 65536x300x300', dict(M=65536, N=300, K=300, tb=1)),
                 ('fs dp NN 65536x128x1000', dict(M=65536, N=128, K=1000)), ('ll dG NT 44467x128x1000', dict(M=44467, N=128, K=1000, tb=1)),
                 ('NN 16384x128x128 tanh', dict(M=16384, N=128, K=128, epi=2)), ('NN 16384x300x300 tanh', dict(M=16384, N=300, K=300, epi=2)),
                 ('NT 32768x256x256', dict(M=32768, N=256, K=256, tb=1)), ('fs logits NT 65536x1000x128', dict(M=65536, N=1000, K=128, tb=1)),
                 ('ll fwd NN 44467x1000x128 bias', dict(M=44467, N=1000, K=128, epi=1)), ('NN 32768x300x300 tanh', dict(M=32768, N=300, K=300, epi=2)), ('NT 32768x128x128', dict(M=32768, N=128, K=128, tb=1)),
                 ('c2 dW TN 128x128x65536 /512', dict(M=128, N=128, K=65536, ta=1, splits=512)), ('c2 dW TN 128x128x65536 /256', dict(M=128, N=128, K=65536, ta=1, splits=256)),
                 ('c4 dW TN 300x300x65536 /113', dict(M=300, N=300, K=65536, ta=1, splits=113)), ('c4 dW TN 300x300x65536 /128', dict(M=300, N=300, K=65536, ta=1, splits=128)),
                 ('ll dW TN 128x1000x44467 /33', dict(M=128, N=1000, K=44467, ta=1, splits=33)), ('ll dW TN 128x1000x44467 /64', dict(M=128, N=1000, K=44467, ta=1, splits=64)),
                 ('fs dRe TN 1000x128x65536 /32', dict(M=1000, N=128, K=65536, ta=1, splits=32)), ('fs dRe TN 1000x128x65536 /64', dict(M=1000, N=128, K=65536, ta=1, splits=64))]:
    us = C.bench_gemm(**kw)
    print('%-46s %8.1f us %7.1f TF' % (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
PY
done
