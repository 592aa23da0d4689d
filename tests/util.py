"""Shared helpers for the parity tests (oracle = checker, HIP engine = subject)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import sert_oracle as O
from sert_amd import _capi as C


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def row_err(a, b, rows=None, floor=1e-3):
    """Row-wise companion of rel_err: max over rows r of ||a_r - b_r||_2 / max(||b_r||_2, floor * median_r ||b_r||_2).
    rel_err is a bound against the largest element of the WHOLE tensor -- a wrong row whose magnitude is 1e-4
    of the largest row passes it; this one prices every row against its own norm (the floor only keeps rows that
    are numerically empty -- cancelled sums, untouched rows under lambda = 0 -- from dividing by ~0).  `rows`
    restricts the maximum to a subset (e.g. the rows a batch touches).  Returns (worst error, its row)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and a.ndim == 2, (a.shape, b.shape)
    if rows is not None:
        rows = np.asarray(rows)
        a, b = a[rows], b[rows]
    nb = np.sqrt((b * b).sum(axis=1))
    den = np.maximum(nb, floor * max(1e-30, float(np.median(nb))))
    e = np.sqrt(((a - b) ** 2).sum(axis=1)) / den
    i = int(np.argmax(e))
    return float(e[i]), (int(rows[i]) if rows is not None else i)


def id_dtype(vocab):
    return np.min_scalar_type(vocab - 1)


def make_vs_problem(seed, N, n, z, Vw, Ve, dw, de, weights='uniform', zipf=False):
    rng = np.random.RandomState(seed)
    Rw = O.glorot_uniform(rng, (Vw, dw))
    Re = O.glorot_uniform(rng, (Ve, de))
    W = O.glorot_uniform(rng, (dw, de))
    b = (0.1 * rng.randn(de)).astype(np.float32)
    if zipf:
        r = np.minimum(rng.zipf(1.1, size=(N, n)) - 1, Vw - 1)
        X = rng.permutation(Vw)[r]
    else:
        X = rng.randint(0, Vw, size=(N, n))
    X = X.astype(id_dtype(Vw))
    y = rng.randint(0, Ve, size=N).astype(np.int32)
    w = (np.ones(N) if weights == 'ones' else rng.uniform(0.5, 2.0, N)).astype(np.float32)
    return dict(Rw=Rw, Re=Re, W=W, b=b, X=X, y=y, w=w, rng=rng)


def vs_engine(p, B, n, z, lam, device=0, keep_grads=1, global_batch=None, seed=1234, **adam):
    Vw, dw = p['Rw'].shape
    Ve, de = p['Re'].shape
    e = C.Engine(kind=C.KIND_VECTORSPACE, batch_size=B, global_batch_size=global_batch or B,
                 window_size=n, vocab_size=Vw, num_entities=Ve, word_dim=dw, entity_dim=de,
                 num_negatives=z, id_bytes=p['X'].dtype.itemsize, device=device,
                 keep_grads=keep_grads, deterministic=1, lambda_=lam,
                 lr=adam.get('lr', 1e-3), beta1=adam.get('beta1', 0.9),
                 beta2=adam.get('beta2', 0.999), eps=adam.get('eps', 1e-8), seed=seed)
    e.set_tensor(C.T_RW, p['Rw'])
    e.set_tensor(C.T_RE, p['Re'])
    e.set_tensor(C.T_W, p['W'])
    e.set_tensor(C.T_B, p['b'])
    return e


def make_ll_problem(seed, N, n, Vw, Ve, d, labels='int'):
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    Rw = O.glorot_uniform(rng, (Vw, d))
    W = O.glorot_uniform(rng, (d, Ve))
    b = (0.1 * rng.randn(Ve)).astype(np.float32)
    X = rng.randint(0, Vw, size=(N, n)).astype(id_dtype(Vw))
    w = rng.uniform(0.5, 2.0, N).astype(np.float32)
    if labels == 'int':
        y = rng.randint(0, Ve, size=N).astype(np.int32)
        ydense = y
    else:
        rows, cols, vals = [], [], []
        for i in range(N):
            k = rng.randint(1, 4)
            idx = np.sort(rng.choice(Ve, k, replace=False))
            rows += [i] * k
            cols += list(idx)
            vals += [1.0 / k] * k
        y = sp.csr_matrix((np.array(vals, dtype=np.float32), (rows, cols)), shape=(N, Ve))
        ydense = np.asarray(y.todense(), dtype=np.float32)
    return dict(Rw=Rw, W=W, b=b, X=X, y=y, ydense=ydense, w=w, rng=rng)


def ll_engine(p, B, n, lam, device=0, keep_grads=1, global_batch=None):
    Vw, d = p['Rw'].shape
    Ve = p['W'].shape[1]
    e = C.Engine(kind=C.KIND_LOGLINEAR, batch_size=B, global_batch_size=global_batch or B,
                 window_size=n, vocab_size=Vw, num_entities=Ve, word_dim=d, entity_dim=0,
                 num_negatives=0, id_bytes=p['X'].dtype.itemsize, device=device,
                 keep_grads=keep_grads, deterministic=1, lambda_=lam,
                 lr=1.0, beta1=0.95, beta2=0.0, eps=1e-6, seed=0)
    e.set_tensor(C.T_RW, p['Rw'])
    e.set_tensor(C.T_W, p['W'])
    e.set_tensor(C.T_B, p['b'])
    return e
