"""gemm_big.h (256x256 tiles, one wave per SIMD) against the production GEMM, NT shapes.

    python tools/bench_gemm_big.py      # on the GPU box; prints both kernels per shape
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [('square 4096', dict(M=4096, N=4096, K=4096)),
          ('scoring 10240x99840x128', dict(M=10240, N=99840, K=128)),
          ('tall 81920x256x1008', dict(M=81920, N=256, K=1008))]

if __name__ == '__main__':
    if len(sys.argv) > 1:
        sys.path.insert(0, ROOT)
        from sert_amd import _capi as C
        for name, kw in SHAPES:
            us = C.bench_gemm(tb=1, iters=5, **kw)
            print('%-12s %-28s %9.1f us %7.1f TF' % (sys.argv[1], name, us, 2.0 * kw['M'] * kw['N'] * kw['K'] / us / 1e6))
    else:
        for tag, env in (('production', {}), ('big 256x256', {'SERT_GEMM_BIG': '1'}), ('mid 256x128', {'SERT_GEMM_BIG': '2'})):
            subprocess.run([sys.executable, os.path.abspath(__file__), tag], check=True, env=dict(os.environ, **env))
