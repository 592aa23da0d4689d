#!/usr/bin/env python
"""Round 3: where do the 13 us per step between `bench.py --steps 20` and `--steps 100` come from?
Host timestamps at every train_fn return of one timed region (the same calls `bench.py:timed_steps`
makes), for a region entered the way the driver's run enters it (5 warm-up steps after the build)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sert_amd import models  # noqa: E402
from sert_amd import distributed  # noqa: E402


def region(model, steps, warmup, num_batches=2):
    eng = model._engine
    for i in range(warmup):
        model.train_fn(i % num_batches)
    eng.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for i in range(steps):
        eng.hint_next_batch((warmup + i + 1) % num_batches if i + 1 < steps else None)
        model.train_fn((warmup + i) % num_batches)
        stamps.append(time.perf_counter() - t0)
    eng.hint_next_batch(None)
    t_hint = time.perf_counter() - t0
    eng.synchronize()
    t_end = time.perf_counter() - t0
    return np.array(stamps) * 1e6, t_hint * 1e6, t_end * 1e6


def main():
    B, n, Vw, Ve, d, z = 65536, 10, 100000, 1000, 128, 10
    if '--c4' in sys.argv:
        Vw, Ve, d = 500000, 100000, 300
    rng = np.random.RandomState(0)
    X, y, w = bench.synth_data(rng, 2 * B, n, Vw, Ve)
    m = bench.build_model('vectorspace', models, B, n, Vw, Ve, d, d, z, X, y, w, seed=0)
    for steps, warmup in ((20, 5), (20, 5), (100, 10), (20, 50)):
        st, th, te = region(m, steps, warmup)
        d_ = np.diff(np.concatenate([[0.0], st]))
        if '--all' in sys.argv:
            print(np.round(d_, 0).astype(int).tolist())
        print('steps %3d warmup %2d: total %.1f us = %.2f us/step; first return %.1f, steps 2-5 %s, median of the rest %.1f, '
              'last return -> hint %.1f -> synchronised %.1f'
              % (steps, warmup, te, te / steps, d_[0], np.round(d_[1:5], 1).tolist(), float(np.median(d_[5:])), th - st[-1], te - th))


if __name__ == '__main__':
    main()
