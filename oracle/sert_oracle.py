"""CPU restatement ("oracle") of SERT's training + entity-scoring hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sert_amd/`` may import this module:
it is the *checker* for the HIP path (``tests/``, ``__graft_entry__.smoke()``)
and the ``cpu_baseline`` leg of ``bench.py``.  The product path fails loudly when
the HIP library is missing; it never falls back to this code.

PARITY STATUS: **parity unpinned for the training arithmetic.**  The reference
executes its arithmetic inside Theano 0.8.2 / Lasagne 0.1 (requirements.txt:3,11),
neither of which is importable here (and neither is vendored under
/root/reference), and the reference ships no tests or golden vectors.  The maths
below restates the reference graph line by line from ``sert/models.py`` plus the
published semantics of the two libraries (tagged [upstream]).  What *is* pinned
against the real reference code (see tests/golden/make_golden.py): the batch
iteration harness, ``sparse_to_one_hot_multiple``, ``WordBatcher`` /
``EmbeddingMapper`` / ``aggregate_distribution`` and both query callbacks.

Every function cites the reference file:line it follows.  ``dtype`` selects the
arithmetic type: float32 mirrors ``floatX=float32`` (product-search.sh:95),
float64 is used by the finite-difference gradient checks.
"""
import numpy as np

EPS = 1e-7  # sert/models.py:200, :290, :900, :1067-1068

# The semantics of Theano 0.8.2 / Lasagne 0.1 this restatement takes FROM MEMORY (neither library is
# importable here; every use is tagged [upstream] below).  Each entry can be flipped to its plausible
# alternative -- tools/semantics_drift.py and tests/test_semantics_cpu.py measure how far a wrong guess
# could move the parameters and the ranking.  The defaults are the restatement.
UPSTREAM = {
    'clip_grad_inclusive': True,    # Clip.grad mask (x >= lo) & (x <= hi); alternative: strict inequalities
    'sigmoid_cutoffs': True,        # ScalarSigmoid float32 c_code: x < -88 -> 0, x > 15 -> 1; alternative: exact
    'sum_acc_float64': True,        # Sum of float32 accumulates in float64; alternative: float32 accumulation
    'adam_eps_outside_sqrt': True,  # m / (sqrt(v) + eps); alternative: m / sqrt(v + eps)
    'adam_folded_bias_correction': True,   # a_t = lr sqrt(1-b2^t)/(1-b1^t) applied to m / (sqrt(v) + eps);
                                           # alternative (Kingma & Ba alg. 1): lr m_hat / (sqrt(v_hat) + eps)
    'bias_regularised': False,      # DenseLayer b has regularizable=False; alternative: b is in the L2 term
    'adadelta_eps_inside_sqrt': True,      # sqrt(delta + eps) / sqrt(accu + eps); alternative: eps outside
}


class upstream_choice(object):
    """with upstream_choice(clip_grad_inclusive=False): ...   -- flip semantics for a block."""

    def __init__(self, **kw):
        unknown = set(kw) - set(UPSTREAM)
        assert not unknown, unknown
        self.kw = kw

    def __enter__(self):
        self.saved = {k: UPSTREAM[k] for k in self.kw}
        UPSTREAM.update(self.kw)

    def __exit__(self, *exc):
        UPSTREAM.update(self.saved)


def _clip_mask(x, lo, hi):
    """Gradient mask of T.clip [upstream: Clip.grad -- inclusive bounds]."""
    if UPSTREAM['clip_grad_inclusive']:
        return (x >= lo) & (x <= hi)
    return (x > lo) & (x < hi)


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #

def glorot_uniform(rng, shape, dtype=np.float32):
    """lasagne.init.GlorotUniform().sample(shape) [upstream: Lasagne 0.1
    init.Glorot: std = sqrt(2/(n1+n2)), Uniform(std) -> a = sqrt(3)*std].
    Called at bin/train.py:128-129, :170-171 and by DenseLayer (models.py:846,
    :1057).  `rng` stands in for the global np.random the reference uses."""
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(low=-a, high=a, size=shape).astype(dtype)


def _sum(x, axis=None, dtype=np.float32):
    """Theano's Sum accumulates float32 inputs in float64 and casts back
    [upstream: CAReduceDtype acc_dtype]."""
    acc = np.float64 if UPSTREAM['sum_acc_float64'] else dtype
    return np.sum(x, axis=axis, dtype=acc).astype(dtype)


def theano_sigmoid(x):
    """T.nnet.sigmoid (models.py:896) [upstream: ScalarSigmoid.c_code for
    float32: x < -88 -> 0, x > 15 -> 1, else 1/(1+exp(-x)); float64 uses
    -709 / 19]."""
    x = np.asarray(x)
    if x.dtype == np.float32:
        lo, hi = np.float32(-88.0), np.float32(15.0)
    else:
        lo, hi = -709.0, 19.0
    one = x.dtype.type(1.0)
    with np.errstate(over='ignore'):
        mid = one / (one + np.exp(-x))
    if not UPSTREAM['sigmoid_cutoffs']:
        return mid.astype(x.dtype)
    return np.where(x < lo, x.dtype.type(0.0),
                    np.where(x > hi, one, mid)).astype(x.dtype)


def softmax_rows(z):
    """T.nnet.softmax (models.py:210, :841) [upstream: row max subtracted,
    exp, divide by the row sum]."""
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m)
    return (e / e.sum(axis=1, keepdims=True)).astype(z.dtype)


def clip_bounds(dtype):
    """The 1e-7 / 1-1e-7 constants as the arithmetic type sees them
    (float32: 1-1e-7 rounds to 1-2**-23)."""
    t = np.dtype(dtype).type
    return t(EPS), t(1.0 - EPS)


# --------------------------------------------------------------------------- #
# optimisers  [upstream: Lasagne 0.1 updates.py]
# --------------------------------------------------------------------------- #

class Adam(object):
    """lasagne.updates.adam, defaults lr=1e-3 b1=.9 b2=.999 eps=1e-8
    (selected at sert/models.py:922, applied :548-549).  One shared step
    counter t (float scalar, starts at 0) for all parameters."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]

    @staticmethod
    def step_size(t, lr, beta1, beta2, dtype):
        """a_t = lr*sqrt(1-b2**t)/(1-b1**t), evaluated in floatX."""
        T = np.dtype(dtype).type
        t = T(t)
        return T(T(lr) * np.sqrt(T(1) - T(beta2) ** t) / (T(1) - T(beta1) ** t))

    def update(self, params, grads):
        self.t += 1
        dt = params[0].dtype
        T = dt.type
        a_t = self.step_size(self.t, self.lr, self.beta1, self.beta2, dt)
        b1, b2, eps = T(self.beta1), T(self.beta2), T(self.eps)
        folded = UPSTREAM['adam_folded_bias_correction']
        eps_out = UPSTREAM['adam_eps_outside_sqrt']
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = b1 * self.m[i] + (T(1) - b1) * g
            self.v[i] = b2 * self.v[i] + (T(1) - b2) * g * g
            if folded:
                den = (np.sqrt(self.v[i]) + eps) if eps_out else np.sqrt(self.v[i] + eps)
                p -= a_t * self.m[i] / den
            else:
                m_hat = self.m[i] / (T(1) - b1 ** T(self.t))
                v_hat = self.v[i] / (T(1) - b2 ** T(self.t))
                den = (np.sqrt(v_hat) + eps) if eps_out else np.sqrt(v_hat + eps)
                p -= T(self.lr) * m_hat / den


class Adadelta(object):
    """lasagne.updates.adadelta, defaults lr=1.0 rho=.95 eps=1e-6
    (selected at sert/models.py:820)."""

    def __init__(self, params, lr=1.0, rho=0.95, eps=1e-6):
        self.lr, self.rho, self.eps = lr, rho, eps
        self.accu = [np.zeros_like(p) for p in params]
        self.delta = [np.zeros_like(p) for p in params]

    def update(self, params, grads):
        T = params[0].dtype.type
        rho, eps, lr = T(self.rho), T(self.eps), T(self.lr)
        for i, (p, g) in enumerate(zip(params, grads)):
            self.accu[i] = rho * self.accu[i] + (T(1) - rho) * g * g
            if UPSTREAM['adadelta_eps_inside_sqrt']:
                upd = g * np.sqrt(self.delta[i] + eps) / np.sqrt(self.accu[i] + eps)
            else:
                upd = g * (np.sqrt(self.delta[i]) + eps) / (np.sqrt(self.accu[i]) + eps)
            p -= lr * upd
            self.delta[i] = rho * self.delta[i] + (T(1) - rho) * upd * upd


# --------------------------------------------------------------------------- #
# vectorspace / LSE  (sert/models.py:905-1118)
# --------------------------------------------------------------------------- #

class VectorSpaceOracle(object):
    """VectorSpaceLanguageModel: window mean-pool -> tanh projection ->
    sigmoid NCE with z uniformly sampled negatives, dense L2, dense Adam.
    Negatives are an explicit input (reference: RandomStreams.choice,
    models.py:956-973 -- iid uniform over entities, with replacement, target
    not excluded)."""

    def __init__(self, batch_size, window_size, num_negative_samples,
                 R_w, R_e, W, b, regularization_lambda, dtype=np.float32,
                 adam_kwargs=None):
        self.B, self.n, self.z = batch_size, window_size, num_negative_samples
        self.dtype = np.dtype(dtype)
        self.lam = regularization_lambda
        c = lambda a: np.array(a, dtype=self.dtype, copy=True)
        self.R_w, self.R_e, self.W, self.b = c(R_w), c(R_e), c(W), c(b)
        # parameter order [R_e, R_w, W, b]: models.py:542-543, :1105
        self.opt = Adam(self.params(), **(adam_kwargs or {}))

    def params(self):
        return [self.R_e, self.R_w, self.W, self.b]

    # -- forward ----------------------------------------------------------- #
    def forward(self, X, y, neg):
        dt = self.dtype
        T = dt.type
        lo, hi = clip_bounds(dt)
        X = np.asarray(X).astype(np.int64)
        G = self.R_w[X]                                     # models.py:180
        h = (_sum(G, axis=1, dtype=dt) / T(self.n)).astype(dt)   # :226
        a = (h @ self.W + self.b).astype(dt)                # :1057 DenseLayer
        t = np.tanh(a)                                      # :1055
        p = np.clip(t, -hi, hi)                             # :1065-1068
        cand = np.concatenate(
            [np.asarray(y, dtype=np.int64)[:, None],
             np.asarray(neg, dtype=np.int64).reshape(len(y), -1)], axis=1)
        E = self.R_e[cand]                                  # :990 (B,1+z,d_e)
        u = _sum(E * p[:, None, :], axis=2, dtype=dt)       # :897-898
        sig = theano_sigmoid(u)                             # :896
        s = np.clip(sig, lo, hi)                            # :900
        logs = np.empty_like(s)
        logs[:, 0] = np.log(s[:, 0])                        # :1091
        logs[:, 1:] = np.log(T(1) - s[:, 1:])               # :1092-1093
        loss = -_sum(logs, axis=1, dtype=dt)                # :1095-1098
        return dict(h=h, a=a, t=t, p=p, cand=cand, u=u, sig=sig, s=s,
                    loss=loss)

    def eval_loss(self, X, y, neg):
        """loss_eval (models.py:751-752): unweighted mean, no regulariser."""
        f = self.forward(X, y, neg)
        return (_sum(f['loss'], dtype=self.dtype) / self.dtype.type(len(y)))

    def regularizer(self):
        """models.py:764-795; b is not regularizable [upstream: Lasagne
        DenseLayer b regularizable=False]."""
        T = self.dtype.type
        if not self.lam > 0.0:
            return T(0)
        m = T(self.B)
        r1 = T(self.lam) * _sum(self.W * self.W, dtype=self.dtype) / (T(2) * m)
        r2 = T(self.lam) * (_sum(self.R_w * self.R_w, dtype=self.dtype) +
                            _sum(self.R_e * self.R_e, dtype=self.dtype)) / (T(2) * m)
        if UPSTREAM['bias_regularised']:
            r1 = r1 + T(self.lam) * _sum(self.b * self.b, dtype=self.dtype) / (T(2) * m)
        return T(r1 + r2)

    # -- loss + gradients -------------------------------------------------- #
    def loss_and_grads(self, X, y, w, neg):
        dt = self.dtype
        T = dt.type
        lo, hi = clip_bounds(dt)
        B = len(y)
        f = self.forward(X, y, neg)
        w = np.asarray(w, dtype=dt)
        loss_train = T(_sum(f['loss'] * w, dtype=dt) / T(B)) + self.regularizer()

        g = (w / T(B)).astype(dt)                          # per-sample scale
        sig, s = f['sig'], f['s']
        mask = _clip_mask(sig, lo, hi).astype(dt)          # Clip.grad inclusive [upstream]
        du = np.empty_like(s)
        # d/du -log(clip(sigma)) = -(1/s) * mask * sigma(1-sigma)
        du[:, 0] = -(g / s[:, 0]) * mask[:, 0] * sig[:, 0] * (T(1) - sig[:, 0])
        # d/du -log(1-clip(sigma)) = (1/(1-s)) * mask * sigma(1-sigma)
        du[:, 1:] = (g[:, None] / (T(1) - s[:, 1:])) * mask[:, 1:] * \
            sig[:, 1:] * (T(1) - sig[:, 1:])
        E = self.R_e[f['cand']]
        dp = _sum(du[:, :, None] * E, axis=1, dtype=dt)
        dR_e = np.zeros_like(self.R_e)
        np.add.at(dR_e, f['cand'].ravel(),
                  (du[:, :, None] * f['p'][:, None, :]).reshape(-1, self.R_e.shape[1]))
        t = f['t']
        da = (dp * _clip_mask(t, -hi, hi).astype(dt) * (T(1) - t * t)).astype(dt)
        dW = (f['h'].T @ da).astype(dt)
        db = _sum(da, axis=0, dtype=dt)
        dh = (da @ self.W.T).astype(dt)
        dR_w = np.zeros_like(self.R_w)
        Xi = np.asarray(X).astype(np.int64)
        np.add.at(dR_w, Xi.ravel(),
                  np.repeat(dh / T(self.n), self.n, axis=0))
        if self.lam > 0.0:
            k = T(self.lam) / T(self.B)
            dW += k * self.W
            dR_w += k * self.R_w
            dR_e += k * self.R_e
            if UPSTREAM['bias_regularised']:
                db = db + k * self.b
        f.update(du=du, dp=dp, da=da, dh=dh)
        return loss_train, [dR_e, dR_w, dW, db], f

    def train_step(self, X, y, w, neg):
        """train_fn (models.py:581-588): returns the loss evaluated BEFORE
        the parameter update."""
        loss, grads, _ = self.loss_and_grads(X, y, w, neg)
        self.opt.update(self.params(), grads)
        return loss

    def predict(self, avg):
        """predict_fn (models.py:1107-1118): tanh(avg.W + b), NO clip."""
        avg = np.asarray(avg, dtype=self.dtype)
        return np.tanh(avg @ self.W + self.b).astype(self.dtype)


class VectorSpaceSoftmaxOracle(VectorSpaceOracle):
    """ADDITIVE variant, not in the reference (BASELINE.json
    configs[1] "embed gather + MFMA projection + full softmax"): the vectorspace
    encoder of models.py:1044-1068 scored against ALL entities,
    logits = clip(t) . R_e^T, loss = clipped categorical cross-entropy
    (clipped_categorical_crossentropy, models.py:289-292), same L2 and Adam.
    Its only pin is this restatement + finite differences."""

    def __init__(self, batch_size, window_size, R_w, R_e, W, b, regularization_lambda,
                 dtype=np.float32, adam_kwargs=None):
        VectorSpaceOracle.__init__(self, batch_size, window_size, 0, R_w, R_e, W, b,
                                   regularization_lambda, dtype, adam_kwargs)

    def forward(self, X, y):
        dt = self.dtype
        T = dt.type
        lo, hi = clip_bounds(dt)
        X = np.asarray(X).astype(np.int64)
        h = (_sum(self.R_w[X], axis=1, dtype=dt) / T(self.n)).astype(dt)
        a = (h @ self.W + self.b).astype(dt)
        t = np.tanh(a)
        p = np.clip(t, -hi, hi)
        Z = (p @ self.R_e.T).astype(dt)
        P = softmax_rows(Z)
        py = P[np.arange(len(y)), np.asarray(y, dtype=np.int64)]
        loss = -np.log(np.clip(py, lo, hi))
        return dict(h=h, a=a, t=t, p=p, Z=Z, P=P, py=py, loss=loss.astype(dt))

    def eval_loss(self, X, y):
        f = self.forward(X, y)
        return _sum(f['loss'], dtype=self.dtype) / self.dtype.type(len(y))

    def loss_and_grads(self, X, y, w):
        dt = self.dtype
        T = dt.type
        lo, hi = clip_bounds(dt)
        B = len(y)
        f = self.forward(X, y)
        w = np.asarray(w, dtype=dt)
        loss_train = T(_sum(f['loss'] * w, dtype=dt) / T(B)) + self.regularizer()
        g = (w / T(B)).astype(dt)
        inside = ((f['py'] >= lo) & (f['py'] <= hi)).astype(dt)
        Y = np.zeros_like(f['P'])
        Y[np.arange(B), np.asarray(y, dtype=np.int64)] = 1
        dZ = ((g * inside)[:, None] * (f['P'] - Y)).astype(dt)
        dR_e = (dZ.T @ f['p']).astype(dt)
        dp = (dZ @ self.R_e).astype(dt)
        t = f['t']
        da = (dp * ((t >= -hi) & (t <= hi)).astype(dt) * (T(1) - t * t)).astype(dt)
        dW = (f['h'].T @ da).astype(dt)
        db = _sum(da, axis=0, dtype=dt)
        dh = (da @ self.W.T).astype(dt)
        dR_w = np.zeros_like(self.R_w)
        np.add.at(dR_w, np.asarray(X).astype(np.int64).ravel(), np.repeat(dh / T(self.n), self.n, axis=0))
        if self.lam > 0.0:
            k = T(self.lam) / T(self.B)
            dW += k * self.W
            dR_w += k * self.R_w
            dR_e += k * self.R_e
        f.update(dZ=dZ, dp=dp, da=da, dh=dh)
        return loss_train, [dR_e, dR_w, dW, db], f

    def train_step(self, X, y, w):
        loss, grads, _ = self.loss_and_grads(X, y, w)
        self.opt.update(self.params(), grads)
        return loss


# --------------------------------------------------------------------------- #
# loglinear  (sert/models.py:804-890)
# --------------------------------------------------------------------------- #

class LogLinearOracle(object):
    """LanguageModel: per-token softmax over all entities, log-product over
    the window, renormalise, clipped categorical cross-entropy, dense L2,
    dense Adadelta.  y is either int labels (B,) or a dense (B,V_e) matrix
    (the densified CSR rows, models.py:66-89)."""

    def __init__(self, batch_size, window_size, R_w, W, b,
                 regularization_lambda, dtype=np.float32, adadelta_kwargs=None):
        self.B, self.n = batch_size, window_size
        self.dtype = np.dtype(dtype)
        self.lam = regularization_lambda
        c = lambda a: np.array(a, dtype=self.dtype, copy=True)
        self.R_w, self.W, self.b = c(R_w), c(W), c(b)
        # parameter order [R_w, W, b]: get_all_params(output_layer), :543
        self.opt = Adadelta(self.params(), **(adadelta_kwargs or {}))

    def params(self):
        return [self.R_w, self.W, self.b]

    def token_distributions(self, X):
        """predict_fn (models.py:880-890): (B,n) ids -> (B,n,V_e)."""
        X = np.asarray(X).astype(np.int64)
        B, n = X.shape
        G = self.R_w[X].reshape(B * n, -1)                  # :180, :838
        Z = (G @ self.W + self.b).astype(self.dtype)        # :846
        P = softmax_rows(Z)                                 # :841
        return G, P.reshape(B, n, -1)                       # :854-856

    def forward(self, X, y):
        dt = self.dtype
        lo, hi = clip_bounds(dt)
        G, P3 = self.token_distributions(X)
        J = _sum(np.log(np.clip(P3, lo, hi)), axis=1, dtype=dt)   # :200-201
        Q = softmax_rows(J)                                 # :210
        Qc = np.clip(Q, lo, hi)                             # :290
        y = np.asarray(y)
        if y.ndim == 1:                                     # :735-737, :292
            loss = -np.log(Qc[np.arange(len(y)), y.astype(np.int64)])
        else:
            loss = -_sum(y.astype(dt) * np.log(Qc), axis=1, dtype=dt)
        return dict(G=G, P3=P3, J=J, Q=Q, Qc=Qc, loss=loss.astype(dt))

    def eval_loss(self, X, y):
        f = self.forward(X, y)
        return _sum(f['loss'], dtype=self.dtype) / self.dtype.type(len(f['loss']))

    def regularizer(self):
        T = self.dtype.type
        if not self.lam > 0.0:
            return T(0)
        m = T(self.B)
        r = T(T(self.lam) * _sum(self.W * self.W, dtype=self.dtype) / (T(2) * m) +
              T(self.lam) * _sum(self.R_w * self.R_w, dtype=self.dtype) / (T(2) * m))
        if UPSTREAM['bias_regularised']:
            r = T(r + T(self.lam) * _sum(self.b * self.b, dtype=self.dtype) / (T(2) * m))
        return r

    def loss_and_grads(self, X, y, w):
        dt = self.dtype
        T = dt.type
        lo, hi = clip_bounds(dt)
        f = self.forward(X, y)
        B, n = np.asarray(X).shape
        w = np.asarray(w, dtype=dt)
        loss_train = T(_sum(f['loss'] * w, dtype=dt) / T(B)) + self.regularizer()
        g = (w / T(B)).astype(dt)
        Q, Qc, P3 = f['Q'], f['Qc'], f['P3']
        y = np.asarray(y)
        if y.ndim == 1:
            Y = np.zeros_like(Q)
            Y[np.arange(B), y.astype(np.int64)] = 1
        else:
            Y = y.astype(dt)
        dQc = -(g[:, None] * Y) / Qc
        dQ = dQc * _clip_mask(Q, lo, hi).astype(dt)
        dJ = Q * (dQ - _sum(dQ * Q, axis=1, dtype=dt)[:, None])
        Pmask = _clip_mask(P3, lo, hi).astype(dt)
        dP = dJ[:, None, :] * Pmask / np.clip(P3, lo, hi)
        dZ = (P3 * (dP - _sum(dP * P3, axis=2, dtype=dt)[:, :, None])).astype(dt)
        dZ2 = dZ.reshape(B * n, -1)
        dW = (f['G'].T @ dZ2).astype(dt)
        db = _sum(dZ2, axis=0, dtype=dt)
        dG = (dZ2 @ self.W.T).astype(dt)
        dR_w = np.zeros_like(self.R_w)
        np.add.at(dR_w, np.asarray(X).astype(np.int64).ravel(), dG)
        if self.lam > 0.0:
            k = T(self.lam) / T(self.B)
            dW += k * self.W
            dR_w += k * self.R_w
            if UPSTREAM['bias_regularised']:
                db = db + k * self.b
        f.update(dJ=dJ, dZ=dZ2, dG=dG)
        return loss_train, [dR_w, dW, db], f

    def train_step(self, X, y, w):
        loss, grads, _ = self.loss_and_grads(X, y, w)
        self.opt.update(self.params(), grads)
        return loss


# --------------------------------------------------------------------------- #
# batch loop  (sert/models.py:351-399, :638-668)
# --------------------------------------------------------------------------- #

def iterate_batches(fn, num_instances, batch_size, shuffle=False, rng=np.random):
    """_iterate_batches: N//B batches, tail dropped (:355-359); the batch
    ORDER is shuffled with np.random.shuffle (:363-367); RuntimeError on a
    non-finite result (:372-379)."""
    num_batches = num_instances // batch_size
    idx = list(range(num_batches))
    if shuffle:
        rng.shuffle(idx)
    results = []
    for i in idx:
        results.append(fn(i))
        if not np.all(np.isfinite(results[-1])):
            raise RuntimeError('Encountered NaN or infinity')
    return num_batches, results


# --------------------------------------------------------------------------- #
# query scoring  (bin/query.py:199-370, sert/inference.py:161-183)
# --------------------------------------------------------------------------- #

def vectorspace_scores(projection, R_e):
    """VectorSpaceCallback: entities L2-normalised (query.py:270-274), the
    query projection L2-normalised (:333-336), score = (<e,p>+1)/2 (:352-357).
    Returns the full (V_e,) score vector for one query."""
    E = R_e / np.linalg.norm(R_e, axis=1)[:, None]
    p = projection.reshape(-1)
    p = p / np.linalg.norm(p)
    return ((E * p[None, :]).sum(axis=1) + 1.0) / 2.0


def vectorspace_rank(projection, R_e, top=None):
    """Candidates = the `top` nearest by euclidean distance on unit vectors
    (== largest cosine) or all (query.py:288-318); sorted by score descending
    (:361-365).  Ties: lowest entity index first (the build's stated rule;
    measure-zero on continuous data)."""
    sc = vectorspace_scores(projection, R_e)
    order = np.argsort(-sc, kind='stable')
    if top is not None and top < len(order):
        order = order[:top]
    return order, sc[order]


def aggregate_product(distribution):
    """inference.aggregate_distribution(mode='product') (inference.py:173-174):
    exp(sum(log P)) with log(0) treated as 0."""
    with np.errstate(divide='ignore'):
        lg = np.where(distribution > 0, np.log(np.where(distribution > 0, distribution, 1)), 0.0)
    return np.exp(lg.sum(axis=0))


def loglinear_rank(token_distributions):
    """LogLinearCallback.process (query.py:209-231): product over the query's
    tokens, renormalise, argsort ascending reversed; ranks ALL entities."""
    d = aggregate_product(np.asarray(token_distributions))
    d = d / d.sum()
    order = np.argsort(d)[::-1]
    return order, d[order]


def ndcg_at_k(ranked_entities, relevant, k=100):
    """Binary-gain nDCG@k (log2 discount), the quantity PRODUCT_SEARCH.md
    quotes from trec_eval; used for the +-1e-4 ranking-parity statement."""
    gains = np.array([1.0 if e in relevant else 0.0 for e in ranked_entities[:k]])
    disc = 1.0 / np.log2(np.arange(2, len(gains) + 2))
    dcg = float((gains * disc).sum())
    ideal = float(disc[:min(len(relevant), k)].sum()) if len(relevant) else 0.0
    return dcg / ideal if ideal > 0 else 0.0
