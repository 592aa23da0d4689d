"""GPU: the mode bench.py TIMES, against the oracle, directly.

The throughput runs draw their negatives on the device (sert/models.py:947-979 semantics: iid uniform, with replacement),
announce the next batch (sert_hint_next_batch: run-ahead of its forward + backward), and -- where a batch touches <= 35 % of
the word table -- update that table lazily with catch-up in registers (kernels_opt.h: dense_update_lazy) and defer the
entity-table update past the step's tail.  Every other oracle test hands the engine explicit negatives, which switches the
run-ahead off; here the oracle is fed the SAME ids the device draws -- oracle/philox.py restates the sampler's stream and is
itself checked against the device (sert_negatives_of_step) -- and the model is driven through ``model.train_fn`` exactly as
``bench.timed_steps`` drives it.  Tolerances: loss 1e-5 relative per step; parameters and Adam moments 1e-4 of
the tensor's largest element, and every word / entity row against its own norm."""
import numpy as np
import pytest

import bench
from oracle import philox
from oracle import sert_oracle as O
from sert_amd import _capi as C
from sert_amd import models
from tests import util as U

pytestmark = pytest.mark.gpu


def _fake_dist():
    class D(object):
        @staticmethod
        def barrier():
            pass

        @staticmethod
        def all_reduce_max(x):
            return x
    return D


def test_oracle_stream_equals_the_device_sampler(hip_lib):
    for B, z, Ve, seed in ((64, 10, 1000, 1234), (33, 7, 5, 99), (4096, 10, 32768, (1 << 30) - 3), (5, 3, 100000, 0)):
        p = U.make_vs_problem(3, B, 2, z, 50, Ve, 8, 8)
        eng = U.vs_engine(p, B, 2, z, 0.01, seed=seed)
        for pos in (0, 1, 7, 123456):
            assert np.array_equal(eng.negatives_of_step(pos), philox.training_negatives(seed, pos, B, z, Ve)), (B, z, Ve, pos)
            assert np.array_equal(eng.negatives_of_step(pos, evaluation=True), philox.evaluation_negatives(seed, pos, B, z, Ve))
        eng.close()


def test_device_sampled_step_equals_the_explicit_one(hip_lib):
    """A step with device-drawn negatives = the same step with sert_negatives_of_step's ids passed explicitly."""
    B, n, z, Vw, Ve, d = 256, 4, 6, 500, 40, 32
    p = U.make_vs_problem(5, 3 * B, n, z, Vw, Ve, d, d)
    outs = []
    for explicit in (False, True):
        eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0, seed=77)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for s in range(5):
            neg = eng.negatives_of_step(eng.get_step()) if explicit else None
            losses.append(eng.train_batch(s % 3, neg))
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_RE).copy()))
        eng.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


CASES = {
    # BASELINE.json configs[1] as bench.py runs it (dense word-table launch: a batch touches 44 % of the rows)
    'c2': dict(B=65536, n=10, Vw=100000, Ve=1000, dw=128, de=128, z=10, nb=4, steps=8, lazy=True),
    # the reference's product-search hyper-parameters (product-search.sh:121-133: batch 4096, d_w 300, d_e 128, z 10) on a
    # 100 k x 32 k vocabulary: lazy word-table update (12 % of the rows per batch), deferred entity-table update
    'product_search': dict(B=4096, n=10, Vw=100000, Ve=32768, dw=300, de=128, z=10, nb=6, steps=8, lazy=True),
    # BASELINE.json configs[3] as bench.py's `c4` record runs it (round-5 verdict, "what's missing" 1): the 150 M-element
    # word table takes dense_update_skip<true, 64, 2>, whose timed form -- hinted: full passes at updates 0, 4 and the
    # unannounced last one, SPARSE launches (rows neither this nor the next batch touches neither read nor written) at
    # 1, 2, 3, 5 -- had only met the HIP-vs-HIP repeat sweep; sorted entity chain, deferred 30 M-element entity update
    'c4': dict(B=65536, n=10, Vw=500000, Ve=100000, dw=300, de=300, z=10, nb=4, steps=7, lazy=True,
               counts=dict(skip_64_2=7, skip_full=3, skip_sparse=4, dense=0, lazy=0), float64=True),
}
# `b` (and `W`) take gradients that are sums over all 65536 batch rows of terms that largely cancel: the float32 oracle's own
# sum is ~3e-4 off its float64 evaluation there (tests/test_gpu_fullbatch.py, same note), and Adam's first steps from b = 0
# turn a relative gradient error into the same relative parameter error.  Where a case asks for it (`float64`) those tensors
# are held to 3e-4 against the float32 oracle and to the 1e-4 of SURVEY 8-d against the SAME oracle run in float64.
DENSE_SUM_TENSORS = ('b', 'W', 'm_W', 'v_W')


@pytest.mark.parametrize('case', sorted(CASES))
def test_timed_mode_against_the_oracle(hip_lib, case):
    c = CASES[case]
    B, n, Vw, Ve, dw, de, z, nb, steps = (c[k] for k in ('B', 'n', 'Vw', 'Ve', 'dw', 'de', 'z', 'nb', 'steps'))
    rng = np.random.RandomState(11)
    X, y, w = bench.synth_data(rng, nb * B, n, Vw, Ve)
    w = rng.uniform(0.5, 2.0, len(w)).astype(np.float32)
    model = bench.build_model('vectorspace', models, B, n, Vw, Ve, dw, de, z, X, y, w, seed=3)
    eng = model._engine
    seed = int(eng.cfg.seed)
    touched = len(np.unique(X[:B])) / float(Vw)
    assert (touched <= 0.5) == c['lazy'], touched     # (sert_hip.hip: lazy up to SERT_LAZY_MAX = 0.5 behind an announcement)
    Rw0, Re0 = eng.get_tensor(C.T_RW).reshape(Vw, dw).copy(), eng.get_tensor(C.T_RE).reshape(Ve, de).copy()
    W0, b0 = eng.get_tensor(C.T_W).reshape(dw, de).copy(), eng.get_tensor(C.T_B).copy()
    ora = O.VectorSpaceOracle(B, n, z, Rw0, Re0, W0, b0, 0.01)
    ora64 = O.VectorSpaceOracle(B, n, z, Rw0, Re0, W0, b0, 0.01, dtype=np.float64) if c.get('float64') else None

    # exactly bench.timed_steps' loop (hint, train_fn, per-step loss read-back), with the losses kept
    order = [(1 + i) % nb for i in range(steps)]
    losses = []
    assert eng.get_step() == 0
    for i, b in enumerate(order):
        eng.hint_next_batch(order[i + 1] if i + 1 < steps else None)
        losses.append(float(model.train_fn(b)))
    eng.hint_next_batch(None)
    eng.synchronize()

    for i, b in enumerate(order):
        sl = slice(b * B, (b + 1) * B)
        neg = philox.training_negatives(seed, i, B, z, Ve)
        ref = float(ora.train_step(X[sl], y[sl], w[sl], neg))
        assert abs(losses[i] - ref) <= 1e-5 * abs(ref), (case, i, losses[i], ref)
        if ora64 is not None:
            ref64 = float(ora64.train_step(X[sl], y[sl], w[sl], neg))
            assert abs(losses[i] - ref64) <= 1e-5 * abs(ref64), (case, i, losses[i], ref64)
    assert eng.get_step() == steps
    # the kernel form that ran (sert_debug_update_counts: host counters of the word-table update's launches)
    for k, v in c.get('counts', {}).items():
        assert eng.update_counts()[k] == v, (case, k, eng.update_counts())

    got = {'R_w': eng.get_tensor(C.T_RW).reshape(Vw, dw), 'R_e': eng.get_tensor(C.T_RE).reshape(Ve, de),
           'W': eng.get_tensor(C.T_W).reshape(dw, de), 'b': eng.get_tensor(C.T_B),
           'm_Rw': eng.get_tensor(C.T_STATE0_RW).reshape(Vw, dw), 'v_Rw': eng.get_tensor(C.T_STATE1_RW).reshape(Vw, dw),
           'm_Re': eng.get_tensor(C.T_STATE0_RE).reshape(Ve, de), 'v_Re': eng.get_tensor(C.T_STATE1_RE).reshape(Ve, de),
           'm_W': eng.get_tensor(C.T_STATE0_W).reshape(dw, de), 'v_W': eng.get_tensor(C.T_STATE1_W).reshape(dw, de)}
    # oracle parameter order [R_e, R_w, W, b] (models.py:542-543, :1105)
    want = {'R_w': ora.R_w, 'R_e': ora.R_e, 'W': ora.W, 'b': ora.b,
            'm_Rw': ora.opt.m[1], 'v_Rw': ora.opt.v[1], 'm_Re': ora.opt.m[0], 'v_Re': ora.opt.v[0],
            'm_W': ora.opt.m[2], 'v_W': ora.opt.v[2]}
    for k in sorted(want):
        tol = 3e-4 if (ora64 is not None and k in DENSE_SUM_TENSORS) else 1e-4
        assert U.rel_err(got[k], want[k]) < tol, (case, k, U.rel_err(got[k], want[k]))
    if ora64 is not None:
        want64 = {'b': ora64.b, 'W': ora64.W, 'm_W': ora64.opt.m[2], 'v_W': ora64.opt.v[2], 'R_w': ora64.R_w, 'R_e': ora64.R_e}
        for k in sorted(want64):
            assert U.rel_err(got[k], want64[k]) < 1e-4, (case, k, 'float64 oracle', U.rel_err(got[k], want64[k]))
    # row-wise: every row against its own norm (a row left one update behind, or updated once too often, shows here and
    # not in the whole-tensor bound); first moments of rows with cancelled sums get the documented floor
    for k in ('R_w', 'R_e', 'm_Rw', 'm_Re', 'v_Rw', 'v_Re'):
        err, row = U.row_err(got[k], want[k])
        assert err < (1e-4 if k in ('R_w', 'R_e') else 2e-3), (case, k, err, row)
    del model


def test_timed_mode_loglinear_at_the_w3c_settings(hip_lib):
    """The reference's W3C expert-finding hyper-parameters (W3C-expert-finding.sh:88-96: loglinear, batch 1024, window 8,
    d = 300; 715 experts -- no multiple of four: the unaligned kernels) as the bench's `small_batch` record drives them:
    hints, run-ahead, lazy Adadelta on the word table, and round 5's trimmed chain (no memset at the head of the step, the
    W / b update on the main stream, row losses straight into the finalisation, word rows stored by the split-K combine of
    dG) -- eight steps against LogLinearOracle (sert/models.py:804-890, :820 restated)."""
    B, n, Vw, Ve, d, nb, steps = 1024, 8, 100000, 715, 300, 6, 8
    rng = np.random.RandomState(17)
    X, y, w = bench.synth_data(rng, nb * B, n, Vw, Ve)
    w = rng.uniform(0.5, 2.0, len(w)).astype(np.float32)
    model = bench.build_model('loglinear', models, B, n, Vw, Ve, d, d, 0, X, y, w, seed=4)
    eng = model._engine
    Rw0 = eng.get_tensor(C.T_RW).reshape(Vw, d).copy()
    W0, b0 = eng.get_tensor(C.T_W).reshape(d, Ve).copy(), eng.get_tensor(C.T_B).copy()
    ora = O.LogLinearOracle(B, n, Rw0, W0, b0, 0.01)
    order = [(2 + i) % nb for i in range(steps)]
    losses = []
    for i, b in enumerate(order):
        eng.hint_next_batch(order[i + 1] if i + 1 < steps else None)
        losses.append(float(model.train_fn(b)))
    eng.hint_next_batch(None)
    eng.synchronize()
    for i, b in enumerate(order):
        sl = slice(b * B, (b + 1) * B)
        ref = float(ora.train_step(X[sl], y[sl], w[sl]))
        assert abs(losses[i] - ref) <= 1e-5 * abs(ref), (i, losses[i], ref)
    got = {'R_w': eng.get_tensor(C.T_RW).reshape(Vw, d), 'W': eng.get_tensor(C.T_W).reshape(d, Ve), 'b': eng.get_tensor(C.T_B),
           'accu_Rw': eng.get_tensor(C.T_STATE0_RW).reshape(Vw, d), 'delta_Rw': eng.get_tensor(C.T_STATE1_RW).reshape(Vw, d),
           'accu_W': eng.get_tensor(C.T_STATE0_W).reshape(d, Ve), 'delta_W': eng.get_tensor(C.T_STATE1_W).reshape(d, Ve)}
    want = {'R_w': ora.R_w, 'W': ora.W, 'b': ora.b, 'accu_Rw': ora.opt.accu[0], 'delta_Rw': ora.opt.delta[0],
            'accu_W': ora.opt.accu[1], 'delta_W': ora.opt.delta[1]}
    for k in sorted(want):
        assert U.rel_err(got[k], want[k]) < 1e-4, (k, U.rel_err(got[k], want[k]))
    err, row = U.row_err(got['R_w'], want['R_w'])
    assert err < 1e-4, (err, row)
    del model


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_in_step_timing_changes_nothing_and_times_the_update(hip_lib, kind):
    """sert_timing_enable(m, 2) (bench.instep_pass: the durations `roofline.frac` is taken on): the normal schedule with every
    plain launch bound to a HIP event pair of its own.  The steps' results are those of untimed steps bit for bit, every group
    the step runs reports a time, the word-table update exactly one launch per step, and more launches than the event ring
    holds (1024) are harvested on the way."""
    B, n, Vw, Ve, d, nb, steps = 2048, 6, 40000, 300, 128, 4, 120
    rng = np.random.RandomState(23)
    X, y, w = bench.synth_data(rng, nb * B, n, Vw, Ve)
    outs = []
    for mode in (0, 2):
        model = bench.build_model(kind, models, B, n, Vw, Ve, d, d, 5, X, y, w, seed=9)
        eng = model._engine
        eng.timing_enable(mode)
        losses = []
        for i in range(steps):
            eng.hint_next_batch((i + 1) % nb if i + 1 < steps else None)
            losses.append(float(model.train_fn(i % nb)))
        eng.synchronize()
        if mode == 2:
            us, launches = eng.timings(), eng.timing_launches()
            assert abs(launches['optimizer_word_table'] - 1.0) < 0.02, launches
            assert 2.0 < us['optimizer_word_table'] < 500.0, us
            # (vectorspace: gather, loss, the word gradient's tree; loglinear: gather of the distinct words, loss, and the
            #  per-word dZ sums, which reuse the entity_grad_reduce slot)
            assert us['gather'] > 0 and us['loss'] > 0 and us['word_grad_segsum' if kind == 'vectorspace' else 'entity_grad_reduce'] > 0, us
            assert sum(launches.values()) * steps > 1024, launches      # (the ring wrapped)
        eng.timing_enable(0)
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_W).copy()))
        del model
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
