"""Query path over the entity-table sizes of the reference's product-search benchmarks
(resources/product-search: 8192 ... 65536 products) and C5's 100 000."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sert_amd import _capi  # noqa: E402

if __name__ == '__main__':
    rng = np.random.RandomState(7)
    Q, d, k = 10000, 128, 100
    P = np.tanh(rng.randn(Q, d)).astype(np.float32)
    for V in (8192, 16384, 32768, 65536, 100000):
        E = rng.randn(V, d).astype(np.float32)
        sc = _capi.Scorer(E)
        sc.topk(P[:256], k)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            sc.topk(P, k)
            best = min(best, time.perf_counter() - t0)
        print('V_e=%6d: %.2f ms, %.2f M queries/s' % (V, 1e3 * best, Q / best / 1e6))
        sc.close()
