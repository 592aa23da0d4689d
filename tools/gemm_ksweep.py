import sys
sys.path.insert(0, '/root/repo')
from sert_amd import _capi as C
for tb in (0, 1):
    for K in (32, 64, 128, 256, 512, 1024):
        us = C.bench_gemm(M=65536, N=128, K=K, tb=tb, epi=0, iters=20)
        print('tb=%d K=%4d  %7.1f us  %6.1f TF' % (tb, K, us, 2.0 * 65536 * 128 * K / us / 1e6))
for K in (128, 1024):
    for epi in (0, 1, 2):
        us = C.bench_gemm(M=65536, N=128, K=K, epi=epi, iters=20)
        print('epi=%d K=%d %7.1f us' % (epi, K, us))
