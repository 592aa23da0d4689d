"""Repeatability of two loglinear steps at C4's table sizes (see r04_ll_c4_ab.py): the same path twice, under the environment
given on the command line."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sert_amd import _capi as C  # noqa: E402
from tests import util as U  # noqa: E402
from oracle import sert_oracle as O  # noqa: E402

B, n, Vw, Ve, d = 1024, 10, 500000, 100000, 300
rng = np.random.RandomState(0)
X, y, w = bench.synth_data(rng, 2 * B, n, Vw, Ve)
p = dict(Rw=O.glorot_uniform(rng, (Vw, d)), W=O.glorot_uniform(rng, (d, Ve)), b=(0.1 * rng.randn(Ve)).astype(np.float32), X=X)


def run():
    eng = U.ll_engine(p, B, n, 0.01, keep_grads=1)
    eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y.astype(np.int32), w=w)
    out = {'loss': [eng.train_batch(0)]}
    for name, t, shape in (('gRw', C.T_GRAD_RW, (Vw, d)), ('gW', C.T_GRAD_W, (d, Ve)), ('gb', C.T_GRAD_B, (1, Ve))):
        out[name] = eng.get_tensor(t, shape).copy()
    out['loss'].append(eng.train_batch(1))
    for name, t, shape in (('gW2', C.T_GRAD_W, (d, Ve)), ('Rw', C.T_RW, (Vw, d)), ('W', C.T_W, (d, Ve)), ('b', C.T_B, (1, Ve))):
        out[name] = eng.get_tensor(t, shape).copy()
    eng.close()
    return out


r1, r2, r3 = run(), run(), run()
for a, b in ((r1, r2), (r2, r3)):
    print({k: float(np.abs(a[k] - b[k]).max() / np.abs(b[k]).max()) for k in ('gRw', 'gW', 'gb', 'gW2', 'Rw', 'W', 'b')})
