"""CPU: the oracle's hand-derived backward passes and optimisers against an INDEPENDENT statement of the
same published maths -- the forward graphs of SURVEY Appendix A written in torch (float64) and
differentiated by torch.autograd, the way Theano's autodiff differentiates the reference's graph
(sert/models.py:542-549), and torch.optim's Adadelta / Adam (the algorithms lasagne.updates implements,
models.py:820, :922).  This is test infrastructure only: torch is not part of the product.

What it can and cannot pin: it pins the oracle's derivative algebra (every parameter at once, clip masks
included) and the Adadelta recurrence.  torch's Adam places epsilon differently from Lasagne 0.1
(m_hat / (sqrt(v_hat) + eps) against a_t m / (sqrt(v) + eps)): the two agree exactly only for eps = 0, which is
what is compared; the epsilon placement itself stays an upstream-memory item (oracle.UPSTREAM,
tools/semantics_drift.py)."""
import numpy as np
import pytest

from oracle import sert_oracle as O

EPS = 1e-7
torch = None      # imported by the tests that use it, NOT at collection: torch brings its own HIP / RCCL
                  # libraries into the process, and the GPU tests of this suite must run against the system ones


@pytest.fixture(autouse=True)
def _lazy_torch():
    global torch
    if torch is None:
        torch = pytest.importorskip('torch')


def _t(a, grad=True):
    return torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float64, requires_grad=grad)


def test_vectorspace_backward_equals_autograd_of_the_forward_graph():
    rng = np.random.RandomState(5)
    B, n, z, Vw, Ve, dw, de, lam = 9, 4, 5, 20, 8, 6, 5, 0.01
    m = O.VectorSpaceOracle(B, n, z, O.glorot_uniform(rng, (Vw, dw), np.float64), O.glorot_uniform(rng, (Ve, de), np.float64),
                            O.glorot_uniform(rng, (dw, de), np.float64), 0.1 * rng.randn(de), lam, np.float64)
    X = rng.randint(0, Vw, (B, n))
    X[0, :] = X[0, 0]                       # duplicate ids inside one window
    y = rng.randint(0, Ve, B)
    w = rng.uniform(.5, 2, B)
    neg = rng.randint(0, Ve, (B, z))
    neg[1, 0] = y[1]                        # a negative that equals the target (models.py:961-973 allows it)
    loss, grads, _ = m.loss_and_grads(X, y, w, neg)

    Rw, Re, W, b = _t(m.R_w), _t(m.R_e), _t(m.W), _t(m.b)
    h = Rw[torch.as_tensor(X)].mean(dim=1)                                   # :180, :226
    p = torch.clamp(torch.tanh(h @ W + b), -1 + EPS, 1 - EPS)                # :1057, :1065-1068
    cand = torch.cat([torch.as_tensor(y)[:, None], torch.as_tensor(neg)], dim=1)
    u = (Re[cand] * p[:, None, :]).sum(-1)                                   # :990, :896-898
    s = torch.clamp(torch.sigmoid(u), EPS, 1 - EPS)                          # :900
    per = -(torch.log(s[:, 0]) + torch.log(1 - s[:, 1:]).sum(1))             # :1091-1098
    reg = lam / (2 * B) * ((W ** 2).sum() + (Rw ** 2).sum() + (Re ** 2).sum())   # :773-793
    L = (torch.as_tensor(w) * per).mean() + reg                              # :278-282
    L.backward()
    assert abs(float(L.detach()) - float(loss)) <= 1e-12 * abs(float(loss))
    named = dict(zip(['R_w', 'R_e', 'W', 'b'], [Rw, Re, W, b]))
    order = [q for q in m.params()]
    for g, q in zip(grads, order):
        ref = [v for k, v in named.items() if v.shape == torch.Size(q.shape) and np.array_equal(v.detach().numpy(), q)][0]
        assert np.abs(g - ref.grad.numpy()).max() <= 1e-12 * max(1e-30, np.abs(ref.grad.numpy()).max())


@pytest.mark.parametrize('labels', ['int', 'dense'])
def test_loglinear_backward_equals_autograd_of_the_forward_graph(labels):
    rng = np.random.RandomState(6)
    B, n, Vw, Ve, d, lam = 7, 4, 15, 9, 5, 0.01
    m = O.LogLinearOracle(B, n, O.glorot_uniform(rng, (Vw, d), np.float64), O.glorot_uniform(rng, (d, Ve), np.float64),
                          0.1 * rng.randn(Ve), lam, np.float64)
    X = rng.randint(0, Vw, (B, n))
    X[2, :] = X[2, 1]
    w = rng.uniform(.5, 2, B)
    if labels == 'int':
        y = rng.randint(0, Ve, B)
        Y = np.eye(Ve)[y]
    else:
        Y = np.zeros((B, Ve))
        for i in range(B):
            k = rng.randint(1, 4)
            Y[i, rng.choice(Ve, k, replace=False)] = 1.0 / k
        y = Y
    loss, grads, _ = m.loss_and_grads(X, y, w)

    Rw, W, b = _t(m.R_w), _t(m.W), _t(m.b)
    G = Rw[torch.as_tensor(X)]                                               # (B, n, d)        :180
    P = torch.softmax(G @ W + b, dim=-1)                                     # :838-849
    J = torch.log(torch.clamp(P, EPS, 1 - EPS)).sum(dim=1)                   # :200-212 (product over the window, in logs)
    Q = torch.softmax(J, dim=-1)
    per = -(torch.as_tensor(Y) * torch.log(torch.clamp(Q, EPS, 1 - EPS))).sum(-1)   # :289-292
    reg = lam / (2 * B) * ((W ** 2).sum() + (Rw ** 2).sum())                 # :773-791
    L = (torch.as_tensor(w) * per).mean() + reg
    L.backward()
    assert abs(float(L.detach()) - float(loss)) <= 1e-12 * abs(float(loss))
    for g, ref in zip(grads, (Rw, W, b)):
        assert np.abs(g - ref.grad.numpy()).max() <= 1e-11 * max(1e-30, np.abs(ref.grad.numpy()).max())


def test_adadelta_equals_torch_adadelta():
    rng = np.random.RandomState(7)
    p0 = rng.randn(50)
    p = [p0.copy()]
    opt = O.Adadelta(p)                                  # lasagne defaults: lr 1.0, rho 0.95, eps 1e-6
    tp = torch.tensor(p0.copy(), dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adadelta([tp], lr=1.0, rho=0.95, eps=1e-6)
    for _ in range(20):
        g = rng.randn(50) * rng.uniform(0.01, 1.0)
        opt.update(p, [g.copy()])
        tp.grad = torch.tensor(g.copy(), dtype=torch.float64)
        topt.step()
        assert np.abs(p[0] - tp.detach().numpy()).max() <= 1e-12 * np.abs(p[0]).max()


def test_adam_bias_correction_equals_torch_adam_at_zero_epsilon():
    rng = np.random.RandomState(8)
    p0 = rng.randn(50)
    p = [p0.copy()]
    opt = O.Adam(p, eps=0.0)
    tp = torch.tensor(p0.copy(), dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adam([tp], lr=1e-3, betas=(0.9, 0.999), eps=0.0)
    for _ in range(20):
        g = rng.randn(50) * rng.uniform(0.01, 1.0)
        opt.update(p, [g.copy()])
        tp.grad = torch.tensor(g.copy(), dtype=torch.float64)
        topt.step()
        assert np.abs(p[0] - tp.detach().numpy()).max() <= 1e-10 * np.abs(p[0]).max()
