"""One epoch the way bin/train.py runs it (train.py:262-348): model construction (data upload +
inverted index), train_error(), validation_error(), train() -- wall times through the Python
surface at the product-search settings (batch 4096, d_w=300, d_e=128, z=10)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == '__main__':
    from sert_amd import models
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    B, n, z, Vw, Ve, dw, de = 4096, 10, 10, 100000, 32768, 300, 128
    rng = np.random.RandomState(0)
    X, y, w = bench.synth_data(rng, N, n, Vw, Ve)
    Xv, yv, _ = bench.synth_data(rng, N // 20, n, Vw, Ve)
    np.random.seed(1)
    r2 = np.random.RandomState(2)
    t0 = time.perf_counter()
    m = models.VectorSpaceLanguageModel(
        batch_size=B, window_size=n, num_negative_samples=z,
        representations_init=bench.glorot(r2, (Vw, dw)), entity_representations_init=bench.glorot(r2, (Ve, de)),
        regularization_lambda=0.01, training_set=(X, y, w), validation_set=(Xv, yv))
    t1 = time.perf_counter()
    te = m.train_error(); t2 = time.perf_counter()
    ve = m.validation_error(); t3 = time.perf_counter()
    nb, loss = m.train(); t4 = time.perf_counter()
    print('N=%d instances, %d batches of %d' % (N, nb, B))
    print('construct (upload + index): %.2f s' % (t1 - t0))
    print('train_error:      %.2f s (%.1f us/batch)  mean %.4f' % (t2 - t1, 1e6 * (t2 - t1) / nb, te[0]))
    print('validation_error: %.2f s  mean %.4f' % (t3 - t2, ve[0]))
    print('train:            %.2f s (%.1f us/batch, %.2f M pairs/s)  mean loss %.4f' % (
        t4 - t3, 1e6 * (t4 - t3) / nb, nb * B / (t4 - t3) / 1e6, loss))
