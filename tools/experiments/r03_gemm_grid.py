#!/usr/bin/env python
"""Round 3: the hot-path GEMM shapes over the persistent grid size (SERT_GEMM_GRID, read once per
process) -- does letting a workgroup walk several tiles pay on the skinny shapes?"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [
    ('c2 fwd', dict(M=65536, N=128, K=128, epi=2)), ('c2 dX', dict(M=65536, N=128, K=128, tb=1)),
    ('c2 dW', dict(M=128, N=128, K=65536, ta=1, splits=512)),
    ('c4 fwd', dict(M=65536, N=300, K=300, epi=2)), ('c4 dX', dict(M=65536, N=300, K=300, tb=1)),
    ('c4 dW', dict(M=300, N=300, K=65536, ta=1, splits=114)),
    ('ll fwd', dict(M=44000, N=1000, K=128, epi=1)), ('ll dX', dict(M=44000, N=128, K=1000, tb=1)),
    ('ll dW', dict(M=128, N=1000, K=44000, ta=1, splits=128)),
]
CODE = '''
import sys; sys.path.insert(0, %r)
from sert_amd import _capi as C
shapes = %r
for name, kw in shapes:
    us = min(C.bench_gemm(iters=30, **kw) for _ in range(2))
    print('%%-8s %%8.1f us %%6.1f TF' %% (name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
'''
for grid in ('256', '512', '768', '1024', '2048'):
    print('--- SERT_GEMM_GRID=%s' % grid)
    sys.stdout.flush()
    subprocess.run([sys.executable, '-c', CODE % (ROOT, SHAPES)], env=dict(os.environ, SERT_GEMM_GRID=grid), cwd=ROOT)
