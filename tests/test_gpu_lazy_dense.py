"""GPU: steps that ALTERNATE between the lazy and the dense form of the word-table update.

The lazy form (kernels_opt.h: dense_update_lazy) runs where a batch touches <= 35 % of the rows, the dense launch
(adam_l2 / adadelta_l2) above that.  A dense step that follows a lazy one with rows left behind must first bring every row
to the number of updates APPLIED so far -- round 4 flushed against the already incremented step counter, which gave every row
one zero-gradient update too many before the dense launch (advisor finding, sert_hip.hip: ensure_rw_current's t_applied).
The data set here mixes batches that touch ~0.4 % of the rows with batches that touch ~45 %, announced by correct hints, so
every transition (lazy -> lazy with rows behind, lazy -> dense, dense -> lazy) is taken; the run must equal BIT FOR BIT
the run without hints (every lazy step writes every row: no row is ever behind) and -- vectorspace -- the keep_grads = 1
run (zeroed gradient table, every row updated in memory every step, never lazy; the loglinear keep_grads = 1 step sends
dG through another GEMM tile form, so its sums agree to rounding only), and the oracle (sert/models.py:548-549, 764-795
restated) within the parity tolerances."""
import numpy as np
import pytest

from oracle import sert_oracle as O
from sert_amd import _capi as C
from tests import util as U

pytestmark = pytest.mark.gpu

B, n, Vw, d = 16384, 10, 270000, 16          # 4.3 M word parameters: above the size below which the table stays dense
SPARSE, DENSE = 's', 'd'
KINDS = [SPARSE, SPARSE, DENSE, SPARSE, DENSE, DENSE, SPARSE, SPARSE, SPARSE, DENSE]      # batch j of the data set


def _tokens(rng, kinds):
    hot = rng.permutation(Vw)[:1000]
    X = np.empty((len(kinds) * B, n), dtype=np.uint32)
    for j, k in enumerate(kinds):
        X[j * B:(j + 1) * B] = hot[rng.randint(0, len(hot), (B, n))] if k == SPARSE else rng.randint(0, Vw, (B, n))
    return X


def _touched(X, j):
    return len(np.unique(X[j * B:(j + 1) * B])) / float(Vw)


@pytest.mark.parametrize('lazy_max', ['0.35', None])
@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_alternating_lazy_and_dense_steps(hip_lib, kind, lazy_max, monkeypatch):
    # SERT_LAZY_MAX=0.35: the ~45 % batches take the dense launch whether announced or not (every transition is taken, and the
    # hinted run equals the unhinted one bit for bit, losses included).  Default (0.5 behind an announcement, 0.35 without):
    # the hinted run is lazy throughout, the unhinted one alternates -- parameters and optimiser state still agree bit for bit,
    # the losses to rounding (dense_update_skip sums p^2 row by row, adam_l2 element by element: kernels_opt.h).
    if lazy_max is not None:
        monkeypatch.setenv('SERT_LAZY_MAX', lazy_max)
    else:
        monkeypatch.delenv('SERT_LAZY_MAX', raising=False)
    same_loss = (lambda a, b: a == b) if lazy_max is not None else (lambda a, b: all(abs(x - y) <= 2e-6 * abs(y) for x, y in zip(a, b)))
    close_loss = lambda a, b: all(abs(x - y) <= 2e-6 * abs(y) for x, y in zip(a, b))
    rng = np.random.RandomState(97)
    nb = len(KINDS)
    if kind == 'vectorspace':
        z, Ve = 4, 12
        p = U.make_vs_problem(97, nb * B, n, z, Vw, Ve, d, d)
        mk = lambda keep: U.vs_engine(p, B, n, z, 0.05, keep_grads=keep)
    else:
        Ve = 24
        p = U.make_ll_problem(97, nb * B, n, Vw, Ve, d, 'int')
        mk = lambda keep: U.ll_engine(p, B, n, 0.05, keep_grads=keep)
    p['X'] = _tokens(rng, KINDS)
    fr = [_touched(p['X'], j) for j in range(nb)]
    assert all((f < 0.01) == (k == SPARSE) and (f > 0.40) == (k == DENSE) for f, k in zip(fr, KINDS)), fr
    order = list(range(nb)) + [1, 2, 0, 4, 3]      # 15 steps: lazy steps with write_all = 0 stand in front of dense ones
    negs = [rng.randint(0, 12, (B, 4)).astype(np.int64) for _ in order] if kind == 'vectorspace' else None
    outs = []
    for keep, hints in ((1, True), (0, False), (0, True)):
        eng = mk(keep)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for s, b in enumerate(order):
            eng.hint_next_batch(order[s + 1] if hints and s + 1 < len(order) else None)
            losses.append(eng.train_batch(b, negs[s]) if negs else eng.train_batch(b))
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_STATE0_RW).copy(),
                     eng.get_tensor(C.T_STATE1_RW).copy(), eng.get_tensor(C.T_W).copy()))
        eng.close()
    assert same_loss(outs[1][0], outs[2][0])
    for a, b_ in zip(outs[1][1:], outs[2][1:]):
        assert np.array_equal(a, b_)
    if kind == 'vectorspace':
        assert close_loss(outs[0][0], outs[2][0])
        for a, b_ in zip(outs[0][1:], outs[2][1:]):
            assert np.array_equal(a, b_)
    else:
        for a, b_ in zip(outs[0][1:], outs[2][1:]):
            assert U.rel_err(a, b_) < 1e-5
    outs = [outs[0], outs[2]]

    # ... and against the oracle (dense update of every row, every step)
    if kind == 'vectorspace':
        ora = O.VectorSpaceOracle(B, n, 4, p['Rw'], p['Re'], p['W'], p['b'], 0.05)
    else:
        ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], 0.05)
    for s, b in enumerate(order):
        sl = slice(b * B, (b + 1) * B)
        ref = ora.train_step(p['X'][sl], p['y'][sl], p['w'][sl], negs[s]) if negs else ora.train_step(p['X'][sl], p['ydense'][sl], p['w'][sl])
        assert abs(outs[1][0][s] - ref) <= 1e-5 * abs(ref), (s, outs[1][0][s], ref)
    Rw = outs[1][1].reshape(Vw, d)
    assert U.rel_err(Rw, ora.R_w) < 1e-4
    err, row = U.row_err(Rw, ora.R_w)
    assert err < 1e-3, (err, row)        # (every row against its own norm: one update too many on an untouched row shows here)
