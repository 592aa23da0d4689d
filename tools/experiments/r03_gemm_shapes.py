#!/usr/bin/env python
"""Round 3: the GEMM shapes of the current steps through sert_bench_gemm (TF of the fp32 MFMA peak 157)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sert_amd import _capi as C  # noqa: E402

for name, kw in [
    ('c2 fwd  65536x128x128 +tanh', dict(M=65536, N=128, K=128, epi=2)), ('c2 dh   NT', dict(M=65536, N=128, K=128, tb=1)),
    ('c2 dW   TN splits 512', dict(M=128, N=128, K=65536, ta=1, splits=512)),
    ('ll fwd  44467x1000x128 +bias', dict(M=44467, N=1000, K=128, epi=1)), ('ll dG   44467x128x1000 NT', dict(M=44467, N=128, K=1000, tb=1)),
    ('ll dW   TN splits 128', dict(M=128, N=1000, K=44467, ta=1, splits=128)),
    ('fs log  65536x1000x128 NT', dict(M=65536, N=1000, K=128, tb=1)), ('fs dp   65536x128x1000', dict(M=65536, N=128, K=1000)),
    ('c4 fwd  65536x300x300 +tanh', dict(M=65536, N=300, K=300, epi=2)), ('c4 dh   NT', dict(M=65536, N=300, K=300, tb=1)),
    ('c4 dW   TN splits 114', dict(M=300, N=300, K=65536, ta=1, splits=114)),
    ('square  4096^3', dict(M=4096, N=4096, K=4096)), ('square  4096^3 NT', dict(M=4096, N=4096, K=4096, tb=1)),
]:
    us = min(C.bench_gemm(iters=20, **kw) for _ in range(3))
    print('%-32s %9.1f us %7.1f TF' % (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
