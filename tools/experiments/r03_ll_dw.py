#!/usr/bin/env python
"""Round 3: the loglinear dW GEMM (G^T.dZ: M = d = 128, N = V_e = 1000, K = distinct words ~ 44 k, split-K) runs at
0.26 of the MFMA peak inside the step while dG and the forward (same operands) reach 0.55: which part of the
shape costs?  sert_bench_gemm over splits / N / the transposed problem."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sert_amd import _capi as C  # noqa: E402

K = 44467
for name, kw in [
    ('dW  128x1000  splits 128', dict(M=128, N=1000, K=K, ta=1, splits=128)),
    ('dW  128x1000  splits 64', dict(M=128, N=1000, K=K, ta=1, splits=64)),
    ('dW  128x1000  splits 32', dict(M=128, N=1000, K=K, ta=1, splits=32)),
    ('dW  128x1000  splits 256', dict(M=128, N=1000, K=K, ta=1, splits=256)),
    ('dW  128x1024  splits 128', dict(M=128, N=1024, K=K, ta=1, splits=128)),
    ('dW^T 1000x128 splits 128', dict(M=1000, N=128, K=K, ta=1, splits=128)),
    ('dW^T 1024x128 splits 128', dict(M=1024, N=128, K=K, ta=1, splits=128)),
    ('NN  128x1000 (A row-major) splits 128', dict(M=128, N=1000, K=K, splits=128)),
    ('dG  44467x128 K=1000 NT', dict(M=K, N=128, K=1000, tb=1)),
    ('fwd 44467x1000 K=128', dict(M=K, N=1000, K=128, epi=1)),
]:
    us = min(C.bench_gemm(iters=20, **kw) for _ in range(2))
    print('%-40s %8.1f us %6.1f TF' % (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
