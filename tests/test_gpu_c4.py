"""-m gpu: BASELINE.json configs[3] ("C4") at its FULL table sizes --
|V_w| = 500 000, |V_e| = 100 000, d_w = d_e = 300, window 10 -- through the C ABI.

  * vectorspace (the reference's LSE model, sert/models.py:1024-1118) at the full batch
    65 536: the known answer (1+z) log 2, bit-determinism run to run, and the
    untouched-row optimiser path against the dense path (bit-identical) over 500 000 rows;
  * vectorspace and loglinear (sert/models.py:804-890) against the oracle with every
    table at full size and a batch the oracle finishes in seconds: the 2-pass entity-key
    sort (17 key bits), the d = 300 column-group paths, the streamed entity-table
    optimiser (4.8 GB of parameters + state per step: the dense L2 + dense update of
    sert/models.py:764-795, :548-549) and the streaming loglinear loss (n V_e floats per
    row do not fit the LDS).

Tolerances as in test_gpu_parity.py (fp32): loss rel 1e-5, parameters rel 1e-4.
"""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from oracle import sert_oracle as O
from sert_amd import _capi as C
from tests import util as U

pytestmark = pytest.mark.gpu

VW, VE, D, N_WIN, Z = 500000, 100000, 300, 10, 10
LOSS_TOL, PARAM_TOL = 1e-5, 1e-4


@pytest.fixture(scope='module')
def c4_tables():
    rng = np.random.RandomState(44)
    return dict(Rw=O.glorot_uniform(rng, (VW, D)), Re=O.glorot_uniform(rng, (VE, D)),
                W=O.glorot_uniform(rng, (D, D)), b=(0.1 * rng.randn(D)).astype(np.float32))


def zipf_tokens(rng, rows, n, vocab):
    ranks = np.minimum(rng.zipf(1.1, size=(rows, n)) - 1, vocab - 1)
    return rng.permutation(vocab).astype(np.uint32)[ranks]


def max_rel(a, b):
    """max |a - b| / max |b| without float64 copies of 600 MB tables."""
    a = np.asarray(a, dtype=np.float32).ravel()
    b = np.asarray(b, dtype=np.float32).ravel()
    return float(np.abs(a - b).max()) / max(1e-30, float(np.abs(b).max()))


def test_c4_vectorspace_oracle_parity_at_full_table_sizes(hip_lib, c4_tables):
    B, steps, lam = 256, 2, 0.01
    rng = np.random.RandomState(5)
    X = zipf_tokens(rng, B * steps, N_WIN, VW)
    y = rng.randint(0, VE, B * steps).astype(np.int32)
    w = rng.uniform(0.5, 2.0, B * steps).astype(np.float32)
    p = dict(c4_tables, X=X)
    eng = U.vs_engine(p, B, N_WIN, Z, lam, keep_grads=0)
    eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y, w=w)
    ora = O.VectorSpaceOracle(B, N_WIN, Z, p['Rw'], p['Re'], p['W'], p['b'], lam)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        neg = rng.randint(0, VE, size=(B, Z)).astype(np.int64)
        ref = ora.train_step(X[sl], y[sl], w[sl], neg)
        got = eng.train_batch(s, neg)
        assert abs(got - ref) <= LOSS_TOL * abs(ref), (s, got, ref)
    assert max_rel(eng.get_tensor(C.T_RW), ora.R_w) < PARAM_TOL
    assert max_rel(eng.get_tensor(C.T_RE), ora.R_e) < PARAM_TOL
    assert max_rel(eng.get_tensor(C.T_W), ora.W) < PARAM_TOL
    assert max_rel(eng.get_tensor(C.T_B), ora.b) < PARAM_TOL
    # the optimiser state of rows no token / label touched still decays (dense update)
    m_rw = eng.get_tensor(C.T_STATE0_RW, (VW, D))
    untouched = np.setdiff1d(np.arange(VW), np.unique(X))[:1000]
    assert np.abs(m_rw[untouched]).max() > 0.0          # the L2 term reached them
    assert max_rel(m_rw[untouched], ora.opt.m[1][untouched]) < PARAM_TOL
    neg = rng.randint(0, VE, size=(B, Z)).astype(np.int64)
    ev, ev_ref = eng.eval_batch(C.SPLIT_TRAIN, 1, neg), ora.eval_loss(X[B:], y[B:], neg)
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    eng.close()


def test_c4_loglinear_oracle_parity_at_full_table_sizes(hip_lib, c4_tables):
    """Streaming loss path (n V_e = 1 M floats per row) + the distinct-word tables, d = 300."""
    B, steps, lam = 48, 2, 0.01
    rng = np.random.RandomState(6)
    X = zipf_tokens(rng, B * steps, N_WIN, VW)
    y = rng.randint(0, VE, B * steps).astype(np.int32)
    w = rng.uniform(0.5, 2.0, B * steps).astype(np.float32)
    Wll = O.glorot_uniform(rng, (D, VE))
    bll = (0.1 * rng.randn(VE)).astype(np.float32)
    p = dict(Rw=c4_tables['Rw'], W=Wll, b=bll, X=X)
    eng = U.ll_engine(p, B, N_WIN, lam, keep_grads=0)
    eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y, w=w)
    ora = O.LogLinearOracle(B, N_WIN, p['Rw'], Wll, bll, lam)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        ref = ora.train_step(X[sl], y[sl], w[sl])
        got = eng.train_batch(s)
        assert abs(got - ref) <= LOSS_TOL * abs(ref), (s, got, ref)
    assert max_rel(eng.get_tensor(C.T_RW), ora.R_w) < PARAM_TOL
    assert max_rel(eng.get_tensor(C.T_W), ora.W) < PARAM_TOL
    assert max_rel(eng.get_tensor(C.T_B), ora.b) < PARAM_TOL
    ev, ev_ref = eng.eval_batch(C.SPLIT_TRAIN, 0), ora.eval_loss(X[:B], y[:B])
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    eng.close()


FULL_BATCH_WORKER = r'''
import sys, zlib, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import sert_oracle as O
from sert_amd import _capi as C
from tests import util as U
VW, VE, D, n, z, B = %(VW)d, %(VE)d, %(D)d, %(n)d, %(z)d, 65536
rng = np.random.RandomState(44)
p = dict(Rw=O.glorot_uniform(rng, (VW, D)), Re=O.glorot_uniform(rng, (VE, D)),
         W=O.glorot_uniform(rng, (D, D)), b=np.zeros(D, np.float32))
ranks = np.minimum(rng.zipf(1.1, size=(2 * B, n)) - 1, VW - 1)
X = rng.permutation(VW).astype(np.uint32)[ranks]
p['X'] = X
y = rng.randint(0, VE, 2 * B).astype(np.int32)
out = {}
# (1) known answer: W = 0, b = 0, lambda = 0  =>  every score is 0, loss = (1+z) log 2
q = dict(p, W=np.zeros((D, D), np.float32))
eng = U.vs_engine(q, B, n, z, 0.0, keep_grads=0, seed=1)
eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y, w=np.ones(2 * B, np.float32))
out['known'] = float(eng.train_batch(0))
eng.close()
# (2) three real steps (device sampler, dense L2, dense Adam over both tables)
eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0, seed=7)
eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y, w=np.ones(2 * B, np.float32))
out['losses'] = [float(eng.train_batch(s %% 2)) for s in range(3)]
for name, which in (('Rw', C.T_RW), ('Re', C.T_RE), ('W', C.T_W), ('b', C.T_B),
                    ('m_Rw', C.T_STATE0_RW), ('v_Re', C.T_STATE1_RE)):
    t = eng.get_tensor(which)
    out['crc_' + name] = zlib.crc32(t.tobytes())
    out['absmax_' + name] = float(np.abs(t).max())
    out['finite_' + name] = bool(np.isfinite(t).all())
eng.close()
print('RESULT ' + json.dumps(out))
'''


def test_c4_vectorspace_full_batch_known_answer_determinism_untouched_rows(hip_lib):
    """B = 65 536 over the full C4 tables, five times in fresh processes: twice as shipped
    (bit-identical: order-fixed reductions everywhere, no float atomics) and once with
    SERT_NO_TOUCHED=1 (every gradient row zeroed and read: the plain dense update) -- the
    untouched-row shortcut must not change a bit over 500 000 rows."""
    import json
    code = FULL_BATCH_WORKER % dict(root=U.ROOT, VW=VW, VE=VE, D=D, n=N_WIN, z=Z)
    outs = []
    for extra in ({}, {}, {'SERT_NO_TOUCHED': '1'}, {'SERT_SIDE_HEAVY': '0'}, {'SERT_RE_DEFER': '0'}):
        r = subprocess.run([sys.executable, '-c', code], check=True, env=dict(os.environ, **extra),
                           cwd=U.ROOT, stdout=subprocess.PIPE, timeout=900)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[len('RESULT '):]))
    a, b, dense, one_queue_opt, no_defer = outs
    assert abs(a['known'] - (1 + Z) * np.log(2.0)) < 2e-5
    assert a == b, 'two identical runs differ'
    assert a == dense, 'untouched-row path differs from the dense path'
    # the shipped schedule for a big entity table (entity chain, dW and the DEFERRED entity-table update on the
    # side stream, its sums of squares carried over from the previous step's launch) against both optimiser
    # launches behind the join, and against the side stream without the deferral: not a bit may differ
    assert a == one_queue_opt, 'side-heavy schedule differs from the plain one'
    assert a == no_defer, 'deferred entity-table update differs from the in-step one'
    assert all(v for k, v in a.items() if k.startswith('finite_'))
    assert a['losses'][2] < a['losses'][0]          # the same batch again, two updates later
    assert a['absmax_m_Rw'] > 0 and a['absmax_v_Re'] > 0


def test_deferred_entity_table_update_between_other_calls(hip_lib):
    """An entity table of more than 2^22 elements takes the side-heavy schedule: its L2 + Adam launch runs
    BEHIND the step's tail and the tail takes the table's sum of squares from the previous step's launch.
    Every other entry point that reads R_e, its state or dR_e has to order itself behind that launch, and a
    host write to the table has to invalidate the carried sums: train / evaluate / read / overwrite / train
    again / several batches per call, each loss and the final tensors against the oracle."""
    B, n, z, Vw, Ve, d, lam = 64, 4, 5, 500, 33000, 128, 0.01
    p = U.make_vs_problem(9, 4 * B, n, z, Vw, Ve, d, d)
    eng = U.vs_engine(p, B, n, z, lam, keep_grads=0)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], lam)
    rng = p['rng']

    def step(s):
        sl = slice(s * B, (s + 1) * B)
        neg = rng.randint(0, Ve, size=(B, z)).astype(np.int64)
        ref = ora.train_step(p['X'][sl], p['y'][sl], p['w'][sl], neg)
        got = eng.train_batch(s, neg)
        assert abs(got - ref) <= LOSS_TOL * abs(ref), (s, got, ref)

    step(0)
    neg = rng.randint(0, Ve, size=(B, z)).astype(np.int64)          # evaluation reads R_e straight behind a step
    ev, ev_ref = eng.eval_batch(C.SPLIT_TRAIN, 1, neg), ora.eval_loss(p['X'][B:2 * B], p['y'][B:2 * B], neg)
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    step(1)
    assert max_rel(eng.get_tensor(C.T_RE), ora.R_e) < PARAM_TOL
    new_re = (ora.R_e * np.float32(1.5)).astype(np.float32)           # the host replaces the table: ||R_e||^2 changes
    eng.set_tensor(C.T_RE, new_re)
    ora.R_e[...] = new_re
    step(2)
    step(3)
    assert max_rel(eng.get_tensor(C.T_STATE1_RE), ora.opt.v[0]) < PARAM_TOL
    for name, which, ref in (('Re', C.T_RE, ora.R_e), ('Rw', C.T_RW, ora.R_w), ('W', C.T_W, ora.W), ('b', C.T_B, ora.b)):
        assert max_rel(eng.get_tensor(which), ref) < PARAM_TOL, name
    eng.close()
