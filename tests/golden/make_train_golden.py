#!/usr/bin/env python
"""Pin the oracle's TRAINING arithmetic against the real reference -- the one command that lifts "parity unpinned".

    THEANO_FLAGS=floatX=float32,device=cpu python tests/golden/make_train_golden.py --reference /path/to/SERT

Needs the reference's 2016 stack: Python 2.7 or 3.5, Theano==0.8.2, Lasagne==0.1 (requirements.txt:3,11), numpy, scipy.
IT CANNOT RUN IN THE BUILD CONTAINER (neither library is installable there) -- that is the point of this file: anyone who
can install the stack runs it once and commits ``tests/golden/train_vectors.npz``; ``tests/test_oracle.py`` then replays the
stored inputs through ``oracle/sert_oracle.py`` and compares every loss, parameter and optimiser-state tensor of every
step.  Until that file exists the oracle's training half is restated from sert/models.py plus the two libraries' published
semantics ([upstream] tags in the oracle) and checked only by finite differences, torch.autograd and the semantic flips
of tests/test_semantics_cpu.py.

What runs is the REAL ``sert.models`` -- no stubs, no mocks of Theano or Lasagne:
  * ``sert.models.VectorSpaceLanguageModel`` and ``sert.models.LanguageModel`` are constructed on tiny seeded problems
    (shapes below) and their compiled ``train_fn`` (models.py:581-588) is called for STEPS batches;
  * the optimiser state is reached by wrapping ``lasagne.updates.adam`` / ``lasagne.updates.adadelta`` (looked up at
    models.py:922 / :820 when the model is constructed) in a pass-through recorder that keeps the OrderedDict of updates the
    real function returns -- its keys are the shared variables (moments, step counter, parameters);
  * the negatives: ``_negative_sampling`` (models.py:947-979) returns ``srng.choice(...)``, a stream seeded from
    ``np.random.randint`` that cannot be replayed outside Theano.  A two-line subclass returns a ``theano.shared`` int64
    matrix instead, set to known ids before every step.  Everything downstream of the ids -- ``T.take`` of the entity rows
    (:990), the sigmoid distance (:893-902), the loss (:1072-1098), autodiff, Adam -- is the reference's graph untouched.

Output (one npz, ~100 kB): for each model `<k>` in {vs, ll_int, ll_csr}: the inputs (`<k>_X`, `_y` or `_y_dense`, `_w`,
initial parameters `_Rw0 _Re0 _W0 _b0`, `_neg` (STEPS, B, z), hyper-parameters `_hp` = [B, n, z, lambda]) and per step s:
`<k>_loss` (STEPS,), `<k>_Rw_s`, `_Re_s`, `_W_s`, `_b_s` and the optimiser state `_s0_<param>_s`, `_s1_<param>_s`
(Adam m, v / Adadelta accu, delta) AFTER step s.
"""
from __future__ import print_function

import argparse
import collections
import os
import sys

import numpy as np

STEPS = 3


def glorot(rng, shape):
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def record_updates(module, name, store):
    real = getattr(module, name)

    def recorder(loss_or_grads, params, *args, **kwargs):
        updates = real(loss_or_grads, params, *args, **kwargs)
        store['params'] = list(params)
        store['updates'] = updates
        return updates
    setattr(module, name, recorder)
    return real


def optimiser_state(store, kind):
    """{param name: (state0 shared, state1 shared)} from the recorded OrderedDict.  Lasagne 0.1 inserts, per parameter,
    adam: m_prev, v_prev, param (then t_prev last); adadelta: accu, param, delta_accu."""
    keys = list(store['updates'].keys())
    out = collections.OrderedDict()
    for p in store['params']:
        i = keys.index(p)
        s0, s1 = (keys[i - 2], keys[i - 1]) if kind == 'adam' else (keys[i - 1], keys[i + 1])
        shape = p.get_value(borrow=True).shape
        assert s0.get_value(borrow=True).shape == shape and s1.get_value(borrow=True).shape == shape, (p, shape)
        out[p] = (s0, s1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', required=True, help='checkout of cvangysel/SERT (its sert/ package is imported from here)')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_vectors.npz'))
    args = ap.parse_args()
    sys.path.insert(0, args.reference)

    import theano
    import lasagne
    import scipy.sparse as sp
    assert theano.config.floatX == 'float32', 'run with THEANO_FLAGS=floatX=float32 (product-search.sh:95)'
    from sert import models          # the real thing

    out = {'theano_version': np.array(theano.__version__), 'lasagne_version': np.array(lasagne.__version__)}

    # ---- vectorspace ------------------------------------------------------------------------------------------------------
    rng = np.random.RandomState(20160721)
    B, n, z, Vw, Ve, dw, de, lam = 8, 3, 4, 40, 9, 6, 5, 0.01
    N = B * STEPS
    X = rng.randint(0, Vw, size=(N, n)).astype(np.min_scalar_type(Vw - 1))
    X[0, :] = X[0, 0]                                   # duplicate ids inside one window
    y = rng.randint(0, Ve, size=N).astype(np.int32)
    w = rng.uniform(0.5, 2.0, N).astype(np.float32)
    Rw0, Re0 = glorot(rng, (Vw, dw)), glorot(rng, (Ve, de))
    neg = rng.randint(0, Ve, size=(STEPS, B, z)).astype(np.int64)
    neg[0, 0, 0] = y[0]                                  # a negative that equals the target (not excluded, models.py:970-973)
    neg[0, 1, :2] = neg[0, 1, 0]                         # a repeated negative (with replacement)
    neg_shared = theano.shared(neg[0], name='forced_negatives')

    class ForcedNegatives(models.VectorSpaceLanguageModel):
        def _negative_sampling(self, num_negative_samples, target_indices):
            assert num_negative_samples == z
            return neg_shared

    store = {}
    real = record_updates(lasagne.updates, 'adam', store)
    np.random.seed(1)
    lasagne.random.set_rng(np.random.RandomState(2))
    empty = (np.zeros((0, n), dtype=X.dtype), np.zeros((0,), dtype=np.int32))
    m = ForcedNegatives(batch_size=B, window_size=n, num_negative_samples=z,
                        representations_init=Rw0.copy(), entity_representations_init=Re0.copy(),
                        regularization_lambda=lam, training_set=(X, y, w), validation_set=empty)
    lasagne.updates.adam = real
    # parameter order [R_e, R_w, W, b]: additional_params + get_all_params(output_layer), models.py:542-543, :1105
    assert [p.get_value().shape for p in store['params']] == [(Ve, de), (Vw, dw), (dw, de), (de,)], store['params']
    byname = collections.OrderedDict(zip(['Re', 'Rw', 'W', 'b'], store['params']))
    state = optimiser_state(store, 'adam')
    out.update(vs_X=X, vs_y=y, vs_w=w, vs_Rw0=Rw0, vs_Re0=Re0, vs_W0=byname['W'].get_value(), vs_b0=byname['b'].get_value(),
               vs_neg=neg, vs_hp=np.array([B, n, z, lam], dtype=np.float64))
    losses = []
    for s in range(STEPS):
        neg_shared.set_value(neg[s])
        losses.append(float(m.train_fn(s)))
        for k, p in byname.items():
            out['vs_%s_%d' % (k, s)] = p.get_value().copy()
            out['vs_s0_%s_%d' % (k, s)] = state[p][0].get_value().copy()
            out['vs_s1_%s_%d' % (k, s)] = state[p][1].get_value().copy()
    out['vs_loss'] = np.array(losses, dtype=np.float64)
    out['vs_test_loss_batch0'] = np.array(float(m.test_fn(0)))      # loss_eval (:751-752) draws from its own stream: forced too
    print('vectorspace losses', losses)

    # ---- loglinear: int labels and CSR labels ----------------------------------------------------------------------------
    for tag, sparse_y in (('ll_int', False), ('ll_csr', True)):
        rng = np.random.RandomState(20160722 + int(sparse_y))
        B, n, Vw, Ve, d, lam = 8, 3, 40, 7, 6, 0.01
        N = B * STEPS
        X = rng.randint(0, Vw, size=(N, n)).astype(np.min_scalar_type(Vw - 1))
        w = rng.uniform(0.5, 2.0, N).astype(np.float32)
        Rw0 = glorot(rng, (Vw, d))
        if sparse_y:
            rows, cols, vals = [], [], []
            for i in range(N):
                k = rng.randint(1, 4)
                idx = np.sort(rng.choice(Ve, k, replace=False))
                rows += [i] * k
                cols += list(idx)
                vals += [1.0 / k] * k
            y = sp.csr_matrix((np.array(vals, dtype=np.float32), (rows, cols)), shape=(N, Ve))
            y_dense = np.asarray(y.todense(), dtype=np.float32)
            empty = (np.zeros((0, n), dtype=X.dtype), sp.csr_matrix((0, Ve), dtype=np.float32))
        else:
            y = rng.randint(0, Ve, size=N).astype(np.int32)
            y_dense = y
            empty = (np.zeros((0, n), dtype=X.dtype), np.zeros((0,), dtype=np.int32))
        store = {}
        real = record_updates(lasagne.updates, 'adadelta', store)
        np.random.seed(3)
        lasagne.random.set_rng(np.random.RandomState(4))
        m = models.LanguageModel(batch_size=B, window_size=n, representations_init=Rw0.copy(), output_layer_size=Ve,
                                 regularization_lambda=lam, training_set=(X, y, w), validation_set=empty)
        lasagne.updates.adadelta = real
        names = ['Rw', 'W', 'b']                        # get_all_params(output_layer): [R_w, W, b], :543
        assert [p.get_value().shape for p in store['params']] == [(Vw, d), (d, Ve), (Ve,)], store['params']
        byname = dict(zip(names, store['params']))
        state = optimiser_state(store, 'adadelta')
        out.update({tag + '_X': X, tag + '_y_dense': y_dense, tag + '_w': w, tag + '_Rw0': Rw0,
                    tag + '_W0': byname['W'].get_value(), tag + '_b0': byname['b'].get_value(),
                    tag + '_hp': np.array([B, n, 0, lam], dtype=np.float64)})
        losses = []
        for s in range(STEPS):
            losses.append(float(m.train_fn(s)))
            for k, p in byname.items():
                out['%s_%s_%d' % (tag, k, s)] = p.get_value().copy()
                out['%s_s0_%s_%d' % (tag, k, s)] = state[p][0].get_value().copy()
                out['%s_s1_%s_%d' % (tag, k, s)] = state[p][1].get_value().copy()
        out[tag + '_loss'] = np.array(losses, dtype=np.float64)
        out[tag + '_test_loss_batch0'] = np.array(float(m.test_fn(0)))
        print(tag, 'losses', losses)

    np.savez_compressed(args.out, **out)
    print('wrote', args.out, '(%d arrays)' % len(out))


if __name__ == '__main__':
    main()
