"""BASELINE configs[0] (C1, the reference's own CPU-runnable case): loglinear, V_w=10k, V_e=100,
d=64, window 5, batch 1024 -- pairs/s through train_batch with the per-step loss read-back."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from sert_amd import _capi as C  # noqa: E402
import util as U  # noqa: E402

if __name__ == '__main__':
    # optional: B n Vw Ve d   (e.g. the W3C expert-finding shape: 1024 8 100000 715 300)
    a = [int(x) for x in sys.argv[1:6]]
    B, n, Vw, Ve, d = (a + [1024, 5, 10000, 100, 64][len(a):])[:5]
    nb = 100 if B * n * 100 <= 2e7 else 20
    for labels in ('int', 'csr'):
        p = U.make_ll_problem(3, B * nb, n, Vw, Ve, d, labels=labels)
        eng = U.ll_engine(p, B, n, 0.01, keep_grads=0)
        if labels == 'int':
            eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        else:
            eng.upload_dataset(C.SPLIT_TRAIN, p['X'], csr=p['y'], w=p['w'])
        for i in range(20):
            eng.train_batch(i % nb)
        t0 = time.perf_counter()
        steps = 500
        for i in range(steps):
            eng.hint_next_batch((i + 1) % nb)
            eng.train_batch(i % nb)
        dt = time.perf_counter() - t0
        print('loglinear B=%d n=%d V_w=%d V_e=%d d=%d (%s labels): %.1f us/step, %.2f M pairs/s' % (B, n, Vw, Ve, d, labels, 1e6 * dt / steps, B * steps / dt / 1e6))
        eng.close()
