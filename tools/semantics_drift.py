#!/usr/bin/env python
"""How much could a wrong guess about Theano 0.8.2 / Lasagne 0.1 move the results?

The training arithmetic of the reference runs inside two libraries that cannot be imported here
(parity unpinned).  oracle/sert_oracle.py restates their semantics from memory and
tags every such choice [upstream] (oracle.UPSTREAM).  This script flips each choice to its plausible
alternative, ONE AT A TIME, trains the same model on the same batches / negatives with the numpy
oracle, and reports against the default restatement:

  * max relative parameter drift after `--steps` steps  (max |p' - p| / max |p| per tensor),
  * the change of the training loss at the last step,
  * |delta nDCG@100| (mean and max over the synthetic query set) and how many top-100 lists differ.

The workload is C2-shaped (BASELINE.json configs[1]: LSE, window 10, z = 10, d = 128, V_e = 1000),
scaled to what numpy finishes in about a minute (--vocab 20000 --batch 4096 by default); labels carry
a learnable structure (entity = f(first token)) so that the ranking means something.

    python tools/semantics_drift.py [--steps 100] [--model vectorspace|loglinear] [--json out.json]

Cites: sert/models.py:893-902 (sigmoid + clip), :1065-1068 (Clip layer), :922 / :820 (Adam / Adadelta),
:92-120 (L2 over regularizable parameters), :200-212 (log-product, softmax).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sert_oracle as O   # noqa: E402

ALTERNATIVES = [
    ('clip_grad_inclusive', False, 'Clip.grad mask with strict inequalities'),
    ('sigmoid_cutoffs', False, 'exact sigmoid instead of the -88 / 15 cut-offs'),
    ('sum_acc_float64', False, 'Sum accumulates in float32'),
    ('adam_eps_outside_sqrt', False, 'Adam: m / sqrt(v + eps)'),
    ('adam_folded_bias_correction', False, 'Adam: lr m_hat / (sqrt(v_hat) + eps) (Kingma & Ba alg. 1)'),
    ('bias_regularised', True, 'projection / output bias inside the L2 term'),
    ('adadelta_eps_inside_sqrt', False, 'Adadelta: eps outside the square roots'),
]


def make_problem(kind, seed, B, nb, n, z, Vw, Ve, d):
    rng = np.random.RandomState(seed)
    ranks = np.minimum(rng.zipf(1.1, size=(nb * B, n)) - 1, Vw - 1)
    X = rng.permutation(Vw)[ranks]
    y = (X[:, 0].astype(np.int64) * 7919 % Ve).astype(np.int32)      # learnable: entity = f(first token)
    w = np.ones(nb * B, np.float32)
    p = dict(X=X, y=y, w=w, Rw=O.glorot_uniform(rng, (Vw, d)))
    if kind == 'vectorspace':
        p.update(Re=O.glorot_uniform(rng, (Ve, d)), W=O.glorot_uniform(rng, (d, d)), b=np.zeros(d, np.float32))
        p['neg'] = rng.randint(0, Ve, size=(64, B, z))
    else:
        p.update(W=O.glorot_uniform(rng, (d, Ve)), b=np.zeros(Ve, np.float32))
    qrng = np.random.RandomState(seed + 1)
    p['queries'] = [X[qrng.randint(0, nb * B), :qrng.randint(1, 7)] for _ in range(200)]
    return p


def train(kind, p, steps, B, nb, n, z, lam, lr):
    if kind == 'vectorspace':
        m = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], lam, adam_kwargs=dict(lr=lr))
    else:
        m = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], lam)
    loss = None
    for s in range(steps):
        j = s % nb
        sl = slice(j * B, (j + 1) * B)
        if kind == 'vectorspace':
            loss = m.train_step(p['X'][sl], p['y'][sl], p['w'][sl], p['neg'][s % len(p['neg'])])
        else:
            loss = m.train_step(p['X'][sl], p['y'][sl], p['w'][sl])
    return m, float(loss)


def rankings(kind, m, p, Ve, k=100):
    out = []
    for q in p['queries']:
        if kind == 'vectorspace':
            proj = m.predict(m.R_w[q].mean(axis=0))
            order, _ = O.vectorspace_rank(proj.astype(np.float64), m.R_e.astype(np.float64), top=k)
        else:
            _, P3 = m.token_distributions(np.asarray(q)[None, :])
            order, _ = O.loglinear_rank(P3[0])
            order = order[:k]
        out.append(np.asarray(order))
    return out


def ndcgs(p, ranks, Ve, k=100):
    vals = []
    for q, order in zip(p['queries'], ranks):
        rel = set(int(int(t) * 7919 % Ve) for t in q)
        vals.append(O.ndcg_at_k(list(order), rel, k))
    return np.array(vals)


def run(kind='vectorspace', steps=100, B=4096, nb=8, n=10, z=10, Vw=20000, Ve=1000, d=128, lam=0.01, lr=1e-3, seed=0,
        verbose=True):
    p = make_problem(kind, seed, B, nb, n, z, Vw, Ve, d)
    t0 = time.time()
    base, base_loss = train(kind, p, steps, B, nb, n, z, lam, lr)
    base_rank = rankings(kind, base, p, Ve)
    base_ndcg = ndcgs(p, base_rank, Ve)
    if verbose:
        print('default restatement: %d steps in %.1f s, last loss %.6f, mean nDCG@100 %.4f'
              % (steps, time.time() - t0, base_loss, base_ndcg.mean()))
    rows = []
    for name, value, text in ALTERNATIVES:
        if (kind == 'vectorspace' and name.startswith('adadelta')) or (kind == 'loglinear' and name.startswith(('adam', 'sigmoid'))):
            continue
        with O.upstream_choice(**{name: value}):
            alt, alt_loss = train(kind, p, steps, B, nb, n, z, lam, lr)
            alt_rank = rankings(kind, alt, p, Ve)
        drift = {}
        for a, b_, tag in zip(alt.params(), base.params(),
                              ('R_e', 'R_w', 'W', 'b') if kind == 'vectorspace' else ('R_w', 'W', 'b')):
            drift[tag] = float(np.abs(a.astype(np.float64) - b_).max() / max(1e-30, np.abs(b_).max()))
        dn = np.abs(ndcgs(p, alt_rank, Ve) - base_ndcg)
        differ = int(sum(not np.array_equal(x, y) for x, y in zip(alt_rank, base_rank)))
        rows.append(dict(choice=name, alternative=text, max_rel_param_drift=max(drift.values()), drift=drift,
                         loss_rel_change=abs(alt_loss - base_loss) / abs(base_loss),
                         ndcg100_mean_abs_delta=float(dn.mean()), ndcg100_max_abs_delta=float(dn.max()),
                         top100_lists_that_differ='%d/%d' % (differ, len(base_rank))))
        if verbose:
            r = rows[-1]
            print('%-30s drift %.2e  loss %.2e  |dnDCG@100| mean %.2e max %.2e  lists %s'
                  % (name, r['max_rel_param_drift'], r['loss_rel_change'], r['ndcg100_mean_abs_delta'],
                     r['ndcg100_max_abs_delta'], r['top100_lists_that_differ']))
    return dict(kind=kind, steps=steps, shape=dict(B=B, n=n, z=z, Vw=Vw, Ve=Ve, d=d, lam=lam, lr=lr),
                base_loss=base_loss, base_ndcg100=float(base_ndcg.mean()), rows=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='both', choices=['vectorspace', 'loglinear', 'both'])
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--vocab', type=int, default=20000)
    ap.add_argument('--entities', type=int, default=1000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    out = []
    for kind in (('vectorspace', 'loglinear') if a.model == 'both' else (a.model,)):
        B = a.batch if kind == 'vectorspace' else min(a.batch, 512)     # loglinear: (B n, V_e) in numpy
        out.append(run(kind, a.steps, B=B, Vw=a.vocab, Ve=a.entities, d=a.dim, lr=a.lr))
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
