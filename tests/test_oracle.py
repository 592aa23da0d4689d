"""CPU: the oracle pinned against analytic facts (the reference ships no tests
and its arithmetic cannot run here: parity of the training maths is otherwise
unpinned -- see oracle/sert_oracle.py header)."""
import numpy as np
import pytest

from oracle import sert_oracle as O


def _fd(params, lossf, eps=1e-6):
    out = []
    for p in params:
        g = np.zeros_like(p)
        it = np.nditer(p, flags=['multi_index'])
        for _ in it:
            i = it.multi_index
            o = p[i]
            p[i] = o + eps
            lp = lossf()
            p[i] = o - eps
            lm = lossf()
            p[i] = o
            g[i] = (lp - lm) / (2 * eps)
        out.append(g)
    return out


def _vs(dtype=np.float64, lam=0.01, seed=0):
    rng = np.random.RandomState(seed)
    B, n, z, Vw, Ve, dw, de = 6, 3, 4, 15, 7, 5, 4
    m = O.VectorSpaceOracle(B, n, z, O.glorot_uniform(rng, (Vw, dw), dtype),
                            O.glorot_uniform(rng, (Ve, de), dtype),
                            O.glorot_uniform(rng, (dw, de), dtype), 0.1 * rng.randn(de), lam, dtype)
    X = rng.randint(0, Vw, (B, n))
    y = rng.randint(0, Ve, B)
    w = rng.uniform(.5, 2, B)
    neg = rng.randint(0, Ve, (B, z))
    return m, X, y, w, neg


def test_vectorspace_gradients_match_finite_differences():
    m, X, y, w, neg = _vs()
    _, grads, _ = m.loss_and_grads(X, y, w, neg)
    num = _fd(m.params(), lambda: m.loss_and_grads(X, y, w, neg)[0])
    for a, b in zip(grads, num):
        assert np.abs(a - b).max() <= 1e-6 * max(1e-12, np.abs(b).max())


@pytest.mark.parametrize('labels', ['int', 'dense'])
def test_loglinear_gradients_match_finite_differences(labels):
    rng = np.random.RandomState(1)
    B, n, Vw, Ve, d = 5, 3, 12, 6, 4
    m = O.LogLinearOracle(B, n, O.glorot_uniform(rng, (Vw, d), np.float64),
                          O.glorot_uniform(rng, (d, Ve), np.float64), 0.1 * rng.randn(Ve), 0.01,
                          np.float64)
    X = rng.randint(0, Vw, (B, n))
    w = rng.uniform(.5, 2, B)
    if labels == 'int':
        y = rng.randint(0, Ve, B)
    else:
        y = np.zeros((B, Ve))
        for i in range(B):
            k = rng.randint(1, 4)
            y[i, rng.choice(Ve, k, replace=False)] = 1.0 / k
    _, grads, _ = m.loss_and_grads(X, y, w)
    num = _fd(m.params(), lambda: m.loss_and_grads(X, y, w)[0])
    for a, b in zip(grads, num):
        assert np.abs(a - b).max() <= 1e-6 * max(1e-12, np.abs(b).max())


def test_known_answers_zero_weights():
    """W=0,b=0: LL loss = log V_e for one-hot y; VS loss = (1+z) log 2."""
    m, X, y, w, neg = _vs(np.float32, lam=0.0)
    m.W[:] = 0
    m.b[:] = 0
    assert abs(m.eval_loss(X, y, neg) - 5 * np.log(2)) < 1e-6
    rng = np.random.RandomState(2)
    ll = O.LogLinearOracle(6, 3, O.glorot_uniform(rng, (15, 5)), np.zeros((5, 7)), np.zeros(7), 0.0)
    assert abs(ll.eval_loss(X, y) - np.log(7)) < 1e-6


def test_duplicate_tokens_accumulate():
    """All tokens equal: the row gradient of that word is sum_i dh_i exactly."""
    m, X, y, w, neg = _vs(lam=0.0)
    X[:] = 3
    _, grads, f = m.loss_and_grads(X, y, w, neg)
    dRw = grads[1]
    assert np.allclose(dRw[3], f['dh'].sum(axis=0), rtol=1e-12, atol=0)
    assert np.abs(np.delete(dRw, 3, axis=0)).max() == 0.0


def test_adam_first_step_moves_by_lr_sign():
    """Step 1 from zero state: |delta| = lr*sqrt(1-b2)/(1-b1) * |g|(1-b1)/(sqrt((1-b2) g^2)+eps) ~ lr."""
    p = [np.array([1.0, -2.0, 3.0])]
    g = [np.array([0.5, -0.25, 0.1])]
    opt = O.Adam(p)
    before = p[0].copy()
    opt.update(p, g)
    assert np.allclose(before - p[0], 1e-3 * np.sign(g[0]), rtol=1e-4)


def test_adadelta_first_step():
    p = [np.array([1.0, -2.0])]
    g = [np.array([0.5, -0.25])]
    opt = O.Adadelta(p)
    before = p[0].copy()
    opt.update(p, g)
    accu = 0.05 * g[0] ** 2
    upd = g[0] * np.sqrt(1e-6) / np.sqrt(accu + 1e-6)
    assert np.allclose(before - p[0], upd)
    assert np.allclose(opt.delta[0], 0.05 * upd ** 2)


def test_clip_saturation_zeroes_gradient():
    """Saturated tanh (|t| > 1-eps): dL/da = 0 for that unit (inclusive clip mask)."""
    m, X, y, w, neg = _vs(np.float32, lam=0.0)
    m.b[0] = 50.0
    _, grads, f = m.loss_and_grads(X, y, w, neg)
    assert np.all(f['da'][:, 0] == 0.0)


def test_train_step_returns_pre_update_loss():
    m, X, y, w, neg = _vs()
    expect, _, _ = m.loss_and_grads(X, y, w, neg)
    got = m.train_step(X, y, w, neg)
    assert got == expect
    assert m.loss_and_grads(X, y, w, neg)[0] != expect


def test_theano_sigmoid_thresholds():
    x = np.array([-100, -88.5, 0, 15.5, 100], dtype=np.float32)
    s = O.theano_sigmoid(x)
    assert s[0] == 0 and s[1] == 0 and s[2] == 0.5 and s[3] == 1 and s[4] == 1


def test_vectorspace_softmax_variant_gradients():
    """The additive full-softmax variant: FD check + W=0 known answer (log V_e)."""
    rng = np.random.RandomState(5)
    B, n, Vw, Ve, dw, de = 6, 3, 15, 7, 5, 4
    m = O.VectorSpaceSoftmaxOracle(B, n, O.glorot_uniform(rng, (Vw, dw), np.float64),
                                   O.glorot_uniform(rng, (Ve, de), np.float64),
                                   O.glorot_uniform(rng, (dw, de), np.float64), 0.1 * rng.randn(de),
                                   0.01, np.float64)
    X = rng.randint(0, Vw, (B, n))
    y = rng.randint(0, Ve, B)
    w = rng.uniform(.5, 2, B)
    _, grads, _ = m.loss_and_grads(X, y, w)
    num = _fd(m.params(), lambda: m.loss_and_grads(X, y, w)[0])
    for a, b in zip(grads, num):
        assert np.abs(a - b).max() <= 1e-6 * max(1e-12, np.abs(b).max())
    m.R_e[:] = 0
    assert abs(m.eval_loss(X, y) - np.log(Ve)) < 1e-9
