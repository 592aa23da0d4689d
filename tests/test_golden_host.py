"""CPU: host logic and the oracle against golden vectors captured from the REAL
reference code (tests/golden/make_golden.py)."""
import importlib.util
import io
import json
import os
import pickle
import types

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import sert_oracle as O
from sert_amd import inference, math_utils, models, scoring

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope='module')
def gold():
    arrays = np.load(os.path.join(HERE, 'golden', 'reference_vectors.npz'))
    with open(os.path.join(HERE, 'golden', 'reference_vectors.json')) as f:
        meta = json.load(f)
    return arrays, meta


def _load_bin(name):
    spec = importlib.util.spec_from_file_location('bin_' + name, os.path.join(ROOT, 'bin', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_iterate_batches_order_and_tail(gold):
    _, meta = gold
    for case in meta['iterate_batches']:
        mi = models.ModelInterface(case['B'])
        visited = []

        def fn(i):
            visited.append(int(i))
            return np.float32(0.5 + i)
        np.random.seed(case['seed'])
        nb, results = mi._iterate_batches(fn, case['N'], shuffle=case['shuffle'])
        assert nb == case['num_batches']
        assert visited == case['visited']
        assert [float(r) for r in results] == case['results']
        # the oracle's restatement of the loop agrees too
        np.random.seed(case['seed'])
        nb2, _ = O.iterate_batches(lambda i: np.float32(1), case['N'], case['B'],
                                   shuffle=case['shuffle'])
        assert nb2 == case['num_batches']


def test_iterate_batches_nonfinite_raises(gold):
    _, meta = gold
    assert meta['iterate_batches_nan_raises'] is True
    mi = models.ModelInterface(4)
    with pytest.raises(RuntimeError):
        mi._iterate_batches(lambda i: np.float32('nan') if i == 1 else np.float32(1), 16)


def test_constants(gold):
    _, meta = gold
    c = meta['constants']
    assert (models.ModelInterface.TRAIN, models.ModelInterface.VALIDATE,
            models.ModelInterface.TEST) == (c['TRAIN'], c['VALIDATE'], c['TEST'])
    assert (inference.WordBatcher.OVERFLOW, inference.WordBatcher.TRUNCATE) == \
        (c['OVERFLOW'], c['TRUNCATE'])


def test_sparse_to_one_hot_multiple(gold):
    arrays, meta = gold
    train = _load_bin('train')
    N, Ve = meta['one_hot_shape']
    y = sp.csr_matrix((arrays['oh_y_data'], arrays['oh_y_indices'], arrays['oh_y_indptr']),
                      shape=(N, Ve))
    new_y, (new_x, new_w) = train.sparse_to_one_hot_multiple(y, arrays['oh_x'], arrays['oh_w'])
    assert new_y.dtype == np.int32 and np.array_equal(new_y, arrays['oh_new_y'])
    assert new_x.dtype == arrays['oh_new_x'].dtype and np.array_equal(new_x, arrays['oh_new_x'])
    assert np.array_equal(new_w, arrays['oh_new_w'])
    ytoy = sp.csr_matrix(np.array([[0, .5, .5], [1, 0, 0], [0, 0, 1]], dtype=np.float32))
    ty, (tx,) = train.sparse_to_one_hot_multiple(ytoy, np.arange(3, dtype=np.int32)[:, None])
    assert [int(v) for v in ty] == meta['one_hot_toy']['y']
    assert [int(v) for v in tx.ravel()] == meta['one_hot_toy']['x']
    # a row without non-zeros is an error
    bad = sp.csr_matrix(np.array([[0, 1.], [0, 0], [1, 0]], dtype=np.float32))
    with pytest.raises(RuntimeError):
        train.sparse_to_one_hot_multiple(bad, np.zeros((3, 1)))


def test_error_delta(gold):
    _, meta = gold
    train = _load_bin('train')
    for case in meta['error_delta']:
        assert list(train.error_delta(case['inp'])) == pytest.approx(case['out'])


def test_train_driver_call_and_dump_sequence(gold, tmp_path):
    _, meta = gold
    train = _load_bin('train')

    class FakeModel(models.ModelInterface):
        def __init__(self, train_errors):
            models.ModelInterface.__init__(self, 8)
            self.calls, self.train_errors, self.k = [], list(train_errors), 0

        def train(self):
            self.calls.append('train')
            return 5, 0.25

        def train_error(self):
            self.calls.append('train_error')
            e = self.train_errors[min(self.k, len(self.train_errors) - 1)]
            self.k += 1
            return e, 0.1

        def validation_error(self):
            self.calls.append('validation_error')
            return 1.0, 0.2

        def get_state(self):
            self.calls.append('get_state')
            return ['PREDICT', np.arange(4, dtype=np.float32), np.arange(6, dtype=np.float32)]

    for ci, case in enumerate(meta['train_driver']):
        fm = FakeModel(case['errors'])
        d = tmp_path / ('run%d' % ci)
        d.mkdir()
        train.train(fm, case['epochs'], str(d / 'model'), abort_threshold=1e-5,
                    early_stopping=False, additional_args=[{'args': 1}])
        assert fm.calls == case['calls']
        files = sorted(os.listdir(str(d)))
        assert files == case['files']
        for f in files:
            c = 0
            with open(str(d / f), 'rb') as fh:
                while True:
                    try:
                        pickle.load(fh)
                        c += 1
                    except EOFError:
                        break
            assert c == case['pickles_per_file'][f]


def test_word_batcher(gold):
    arrays, meta = gold
    wbm = meta['wb']
    table = arrays['wb_table']
    batches, calls = [], []

    def predict_fn(batch, mask):
        batches.append((batch.copy(), mask.copy()))
        return table[batch.astype(np.int64)]

    class CB(object):
        def __call__(self, payload, result, **kw):
            calls.append((list(payload), np.array(result), dict(kw)))

        def should_average_input(self):
            return False
    wb = inference.create(predict_fn, None, wbm['B'], wbm['n'], wbm['Vw'], CB())
    assert str(wb.batch.dtype) == wbm['dtype']
    for qi, q in enumerate(wbm['queries']):
        wb.submit(list(q), topic_id='t%d' % qi)
    wb.process()
    assert len(batches) == wbm['num_batches']
    for bi, (b, m) in enumerate(batches):
        assert np.array_equal(b, arrays['wb_batch_%d' % bi])
        assert np.array_equal(m, arrays['wb_mask_%d' % bi]) and m.dtype == np.int8
    assert [c[0] for c in calls] == wbm['call_payloads']
    assert [c[2]['topic_id'] for c in calls] == wbm['call_topics']
    for ci, c in enumerate(calls):
        assert np.array_equal(c[1], arrays['wb_result_%d' % ci])
    assert meta['wb_overlong_raises'] is True
    wb2 = inference.create(predict_fn, None, 2, 3, wbm['Vw'], CB())
    with pytest.raises(RuntimeError):
        wb2.submit(list(range(7)), topic_id='x')


def test_embedding_mapper(gold):
    arrays, _ = gold
    seen, calls = [], []

    class CB(object):
        def __call__(self, payload, result, **kw):
            calls.append((payload, result, kw))

        def should_average_input(self):
            return True
    em = inference.create(lambda avg: seen.append(np.array(avg)) or avg[None, :] * 2.0,
                          arrays['em_Rw'], 4, 3, 50, CB())
    em.submit([3, 4, 10], topic_id='q')
    em.process()
    assert np.array_equal(seen[0], arrays['em_avg'])
    assert np.array_equal(calls[0][1], arrays['em_result'])
    assert calls[0][2] == {'topic_id': 'q'}


def test_aggregate_distribution_and_entropy(gold):
    arrays, meta = gold
    D = arrays['agg_in']
    for mode in ['sum', 'product', 'last', 'max', 'identity']:
        assert np.array_equal(inference.aggregate_distribution(D, mode, 0), arrays['agg_' + mode])
    assert np.array_equal(inference.aggregate_distribution(np.array([[0, .5], [.5, .5]]), 'product', 0),
                          arrays['agg_zero_case'])
    assert np.allclose(O.aggregate_product(D), arrays['agg_product'], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        inference.aggregate_distribution(D, 'nope', 0)
    e = arrays['entropy_in']
    assert math_utils.entropy(e) == pytest.approx(meta['entropy']['plain'])
    assert math_utils.entropy(e, base=2, normalize=True) == pytest.approx(meta['entropy']['base2_norm'])


def test_loglinear_callback_matches_reference(gold):
    arrays, meta = gold
    for qi in range(meta['ll_num']):
        P = arrays['ll_in_%d' % qi]
        ranked = []
        cb = scoring.LogLinearCallback(types.SimpleNamespace(), types.SimpleNamespace(),
                                       {i: 'w%d' % i for i in range(10)}, io.StringIO(),
                                       lambda t, idx, val: ranked.append((np.array(idx), np.array(val))))
        cb(list(range(P.shape[0])), P.copy(), topic_id='q')
        assert np.array_equal(ranked[0][0], arrays['ll_idx_%d' % qi])
        assert np.allclose(ranked[0][1], arrays['ll_val_%d' % qi], rtol=1e-6)
        assert np.allclose(scoring.compute_normalised_entropy(P, base=2), arrays['ll_entropies_%d' % qi])
        # oracle restatement of the same ranking
        order, vals = O.loglinear_rank(P)
        assert np.array_equal(order, arrays['ll_idx_%d' % qi])
        assert np.allclose(vals, arrays['ll_val_%d' % qi], rtol=1e-5)
    # a topic may only be scored once (query.py:182)
    with pytest.raises(AssertionError):
        cb(list(range(P.shape[0])), P.copy(), topic_id='q')


@pytest.mark.parametrize('tag,top', [('10', 10), ('all', None), ('100', None)])
def test_oracle_vectorspace_rank_matches_reference(gold, tag, top):
    """The oracle's scoring restatement vs the reference's VectorSpaceCallback
    (sklearn brute kNN / cdist + Python candidate loop)."""
    arrays, _ = gold
    E, projs = arrays['vs_E'], arrays['vs_proj']
    for qi in range(projs.shape[0]):
        idx = arrays['vs_top%s_idx_%d' % (tag, qi)]
        val = arrays['vs_top%s_val_%d' % (tag, qi)]
        order, sc = O.vectorspace_rank(projs[qi].astype(np.float64), E.astype(np.float64), top=top)
        assert len(order) == len(idx)
        full = O.vectorspace_scores(projs[qi].astype(np.float64), E.astype(np.float64))
        for r in np.nonzero(order != idx)[0]:
            assert abs(full[idx[r]] - sc[r]) < 1e-6      # only near-ties may swap
        assert np.abs(sc - val).max() < 1e-6


def test_epoch_loop_announces_the_following_batch():
    """_iterate_batches tells the engine which batch follows (sert_hint_next_batch) when it
    drives train_fn -- and only then; order and results are those of the reference loop."""
    from sert_amd import models

    class FakeEngine(object):
        def __init__(self):
            self.hints, self.trained = [], []

        def hint_next_batch(self, j):
            self.hints.append(j)

    class M(models.ModelInterface):
        def __init__(self):
            self.batch_size = 4
            self._engine = FakeEngine()

        def train_fn(self, j):
            self._engine.trained.append(j)
            return np.float32(j)

        def test_fn(self, j):
            return np.float32(-j)

    m = M()
    np.random.seed(3)
    nb, res = m._iterate_batches(m.train_fn, 4 * 5 + 2, shuffle=True)
    assert nb == 5 and [int(r) for r in res] == m._engine.trained
    assert m._engine.hints == m._engine.trained[1:] + [None]
    m._engine.hints = []
    m._iterate_batches(m.test_fn, 4 * 5)
    assert m._engine.hints == []


def test_instances_to_arrays_matches_reference():
    """bin/prepare.py:543-599 (instances_and_labels_to_arrays), run from the real reference
    by tests/golden/make_golden.py: dense id matrix, CSR label matrix through a non-identity
    class mapping, np.random shuffle order under a seed."""
    import scipy.sparse as sp
    from sert_amd import prepare as prep
    with open(os.path.join(HERE, 'golden', 'reference_vectors.json')) as f:
        meta = json.load(f)
    assert len(meta['instances_to_arrays']) == 3
    for case in meta['instances_to_arrays']:
        instances = [(d, tuple(w), dict(l)) for d, w, l in case['instances']]
        np.random.seed(case['seed'])
        x, y = prep.to_arrays(instances, case['window_size'], case['class_mapping'],
                              np.dtype(case['x_dtype']), case['shuffle'])
        assert x.dtype == np.dtype(case['x_dtype'])
        assert x.tolist() == case['x']
        y = sp.csr_matrix(y)
        y.sort_indices()
        assert list(y.shape) == case['y_shape']
        assert y.indptr.tolist() == case['y_indptr']
        assert y.indices.tolist() == case['y_indices']
        assert [float(v) for v in y.data] == case['y_data']


def test_bf16_prefilter_error_bound():
    """kernels_score_bf16.h: |<bf16(a), bf16(b)> - <a, b>| <= kBf16Delta = 0.0079 whenever
    |a|, |b| <= 1 (bf16 keeps 8 significant bits: round-to-nearest error <= 2^-8 relative per
    operand, products exact in fp32).  The bound is attained to within 1 % by a = b with every
    component just below a rounding boundary, and random unit vectors stay far inside it."""
    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
        return u.astype(np.uint32).view(np.float32)
    delta = 0.0079
    # worst case: v = 2^-4 (1 + 2^-8 - 2^-20) rounds down to 2^-4; 254 copies have norm < 1
    v = np.float32(2.0 ** -4 * (1 + 2.0 ** -8 - 2.0 ** -20))
    a = np.full(254, v, np.float32)
    assert float(np.linalg.norm(a.astype(np.float64))) <= 1.0 and bf16(a)[0] == np.float32(2.0 ** -4)
    err = abs(float(np.dot(bf16(a).astype(np.float64), bf16(a).astype(np.float64))) -
              float(np.dot(a.astype(np.float64), a.astype(np.float64))))
    assert 0.99 * 2.0 ** -7 < err <= delta
    rng = np.random.RandomState(5)
    for d in (32, 128, 300):
        x = rng.randn(4000, d); y = rng.randn(4000, d)
        x /= np.linalg.norm(x, axis=1, keepdims=True); y /= np.linalg.norm(y, axis=1, keepdims=True)
        x32, y32 = x.astype(np.float32), y.astype(np.float32)
        exact = np.einsum('ij,ij->i', x32.astype(np.float64), y32.astype(np.float64))
        approx = np.einsum('ij,ij->i', bf16(x32).astype(np.float64), bf16(y32).astype(np.float64))
        assert float(np.abs(approx - exact).max()) < delta / 4
