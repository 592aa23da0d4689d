#!/usr/bin/env python
"""GEMM micro-benchmark over the shapes of the hot path (run on the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sert_amd import _capi as C

SHAPES = [
    ('vs fwd  h.W tanh      ', dict(M=65536, N=128, K=128, epi=2)),
    ('vs dh   da.W^T        ', dict(M=65536, N=128, K=128, tb=1)),
    ('vs dW   h^T.da splitK ', dict(M=128, N=128, K=65536, ta=1, splits=256)),
    ('ll fwd  G.W+b         ', dict(M=81920, N=1000, K=128, epi=1)),
    ('ll dG   dZ.W^T        ', dict(M=81920, N=128, K=1000, tb=1)),
    ('ll dW   G^T.dZ splitK ', dict(M=128, N=1000, K=81920, ta=1, splits=128)),
    ('query   P.E^T         ', dict(M=10000, N=100000, K=128, tb=1, iters=3)),
    ('square 4096 NN        ', dict(M=4096, N=4096, K=4096, iters=5)),
    ('square 4096 NT        ', dict(M=4096, N=4096, K=4096, tb=1, iters=5)),
]
for name, kw in SHAPES:
    us = C.bench_gemm(**kw)
    fl = 2.0 * kw['M'] * kw['N'] * kw['K']
    print('%s %10.1f us  %7.1f TFLOP/s' % (name, us, fl / us / 1e6))
