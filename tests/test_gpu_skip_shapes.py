"""GPU: every template shape of dense_update_skip in front of the dense launch AND the oracle.

The lazy word-table update that does not read the rows nobody needs (kernels_opt.h: dense_update_skip<ADAM, lanes per row,
float4 columns per lane>) is instantiated in six shapes by the row width (sert_hip.hip: SERT_SKIP_SHAPE) and twice by the
optimiser -- Adam for the vectorspace model (sert/models.py:922), Adadelta for the loglinear one (:820), both applied by
:548-549 to the L2-regularised gradient of :764-795.  Round 5's tests launched <32,1> (C2) and <32,3> (the product-search
settings) only; `--word_representation_size 200` on a 21 k vocabulary takes <64,1>, C4 takes <64,2>, wider rows <64,3> and
<64,4>.  Here every shape runs the hint plan of test_lazy_word_table_update_is_bit_exact (right hints: rows stay behind for up
to three updates; wrong ones: the forward finds its rows stale; none: everything written; a tensor read and a step-counter
write in between) and must
  * equal the dense run (keep_grads = 1: every row updated in memory every step) BIT FOR BIT in the table and both moments,
  * equal the oracle (loss 1e-5 per step, table 1e-4 of its maximum and every row against its own norm),
  * and have LAUNCHED the shape it names, full passes and sparse ones both (sert_debug_update_counts)."""
import numpy as np
import pytest

from oracle import philox
from oracle import sert_oracle as O
from sert_amd import _capi as C
from tests import util as U

pytestmark = pytest.mark.gpu

# (d_w, V_w, the counter that must move): V_w . d_w >= 2^22 elements (below that the table stays dense); d_w = 300 twice:
# below 2^26 elements the 32 x 3 form, above it 64 x 2 (sert_hip.hip: skip_32x3)
SHAPES = [
    (64, 66000, 'skip_32_1'), (128, 33000, 'skip_32_1'),
    (132, 32000, 'skip_64_1'), (200, 21000, 'skip_64_1'), (256, 16500, 'skip_64_1'),
    (300, 14000, 'skip_32_3'), (384, 11000, 'skip_32_3'),
    (300, 224000, 'skip_64_2'), (400, 10500, 'skip_64_2'), (512, 8200, 'skip_64_2'),
    (516, 8200, 'skip_64_3'), (768, 5500, 'skip_64_3'),
    (772, 5500, 'skip_64_4'), (1024, 4100, 'skip_64_4'),
]
# per step: (batch, hint given before the step) -- None: no hint
PLAN = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0), (0, 3), (1, None), (2, 3), (3, 4), (4, 0), (0, 1), (1, 2), (2, 0), (0, None)]


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
@pytest.mark.parametrize('shape', SHAPES, ids=['d%d_%s%s' % (s[0], s[2], '_big' if s[1] * s[0] >= 1 << 26 else '') for s in SHAPES])
def test_every_skip_shape_against_the_dense_launch_and_the_oracle(hip_lib, kind, shape):
    d, Vw, counter = shape
    assert Vw * d >= 1 << 22
    B, n, seed = 32, 3, 4321
    if kind == 'vectorspace':
        z, Ve, de = 4, 12, 16
        p = U.make_vs_problem(43, B * 5, n, z, Vw, Ve, d, de)
        mk = lambda keep: U.vs_engine(p, B, n, z, 0.05, keep_grads=keep, seed=seed)
    else:
        Ve = 24
        p = U.make_ll_problem(43, B * 5, n, Vw, Ve, d, 'int')
        mk = lambda keep: U.ll_engine(p, B, n, 0.05, keep_grads=keep)
    outs = []
    for keep in (1, 0):
        eng = mk(keep)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for s, (b, hint) in enumerate(PLAN):
            if hint is not None:
                eng.hint_next_batch(hint)
            losses.append(eng.train_batch(b))
            if s == 8:
                float(eng.get_tensor(C.T_RW).sum())         # (a tensor read flushes the rows that are behind)
            if s == 10:
                eng.set_step(eng.get_step())                # (so does a write of the step counter)
        counts = eng.update_counts()
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_STATE0_RW).copy(), eng.get_tensor(C.T_STATE1_RW).copy(), counts))
        eng.close()
    # the forms that ran: keep_grads = 1 the dense launch every step; the product configuration the named shape of
    # dense_update_skip, as passes that read everything and as sparse ones
    assert outs[0][4]['dense'] == len(PLAN) and sum(outs[0][4][k] for k in outs[0][4] if k != 'dense') == 0, outs[0][4]
    c = outs[1][4]
    assert c[counter] == len(PLAN) and c['dense'] == 0 and c['lazy'] == 0, c
    assert c['skip_sparse'] >= 5 and c['skip_full'] >= 4 and c['skip_sparse'] + c['skip_full'] == len(PLAN), c
    # bit for bit against the dense run (the losses: the same squares through another summation tree, kernels_opt.h)
    assert all(abs(x - y) <= 2e-6 * abs(y) for x, y in zip(outs[0][0], outs[1][0])), (outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:4], outs[1][1:4]):
        assert np.array_equal(a, b)
    # ... and against the oracle (dense update of every row, every step)
    if kind == 'vectorspace':
        ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.05)
    else:
        ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], 0.05)
    for s, (b, _) in enumerate(PLAN):
        sl = slice(b * B, (b + 1) * B)
        if kind == 'vectorspace':
            ref = ora.train_step(p['X'][sl], p['y'][sl], p['w'][sl], philox.training_negatives(seed, s, B, z, Ve))
        else:
            ref = ora.train_step(p['X'][sl], p['ydense'][sl], p['w'][sl])
        assert abs(outs[1][0][s] - ref) <= 1e-5 * abs(ref), (s, outs[1][0][s], ref)
    Rw = outs[1][1].reshape(Vw, d)
    assert U.rel_err(Rw, ora.R_w) < 1e-4
    err, row = U.row_err(Rw, ora.R_w)
    assert err < 1e-4, (err, row)        # (every row against its own norm: one update too many on an untouched row shows here)
    k = 1 if kind == 'vectorspace' else 0          # oracle parameter order [R_e, R_w, W, b] / [R_w, W, b]
    s0, s1 = (ora.opt.m[k], ora.opt.v[k]) if kind == 'vectorspace' else (ora.opt.accu[k], ora.opt.delta[k])
    assert U.rel_err(outs[1][2].reshape(Vw, d), s0) < 1e-4 and U.rel_err(outs[1][3].reshape(Vw, d), s1) < 1e-4
