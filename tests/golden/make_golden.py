#!/usr/bin/env python
"""Generate golden vectors from the REAL reference code (run in the build
container only: /root/reference does not exist on the GPU box).

The reference's arithmetic lives in Theano/Lasagne, which cannot be imported
here; what CAN be imported is every pure-NumPy piece around it:

  * sert.inference           (as is)
  * sert.math_utils          (as is)
  * sert.models              with MagicMock stubs for theano / lasagne:
                             ModelInterface._iterate_batches, ModelBase.train /
                             train_error / validation_error / get_state
  * bin/train.py             sparse_to_one_hot_multiple, error_delta, train()
  * bin/query.py             VectorSpaceCallback, LogLinearCallback,
                             compute_normalised_entropy

This script feeds seeded inputs through those functions and stores inputs and
outputs in tests/golden/reference_vectors.npz (+ .json for structured data).
Only DATA is written; no reference source is copied.

    python tests/golden/make_golden.py
"""
import importlib.util
import io
import json
import os
import pickle
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import scipy.sparse as sp

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ['theano', 'theano.tensor', 'theano.sparse', 'theano.compile',
                 'theano.compile.nanguardmode', 'theano.tensor.shared_randomstreams',
                 'lasagne', 'lasagne.layers', 'lasagne.init', 'lasagne.updates',
                 'lasagne.nonlinearities', 'lasagne.objectives',
                 'cvangysel']:
        sys.modules[name] = mock.MagicMock()
    sys.modules['lasagne'].layers.Layer = type('Layer', (object,), {})
    sys.modules['lasagne.layers'].Layer = sys.modules['lasagne'].layers.Layer
    cv = sys.modules['cvangysel']
    cv.sklearn_utils.neighbors_algorithm = lambda metric: 'brute'
    sys.path.insert(0, REF)
    import sert.inference as inference
    import sert.math_utils as math_utils
    import sert.models as models

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    train = load(os.path.join(REF, 'bin', 'train.py'), 'ref_train')
    query = load(os.path.join(REF, 'bin', 'query.py'), 'ref_query')
    # bin/prepare.py: nltk and the pre-0.18 sklearn.cross_validation module are absent too
    for name in ['nltk', 'nltk.corpus', 'sklearn.cross_validation']:
        sys.modules[name] = mock.MagicMock()
    global prepare_mod
    prepare_mod = load(os.path.join(REF, 'bin', 'prepare.py'), 'ref_prepare')
    return inference, math_utils, models, train, query


prepare_mod = None


def main():
    inference, math_utils, models, train, query = import_reference()
    arrays, meta = {}, {}

    # ---- f1: instances_and_labels_to_arrays (prepare.py:543-599) ---------------
    rng = np.random.RandomState(17)
    ents = ['E%02d' % i for i in range(7)]
    class_mapping = {e: i for i, e in enumerate(sorted(ents, reverse=True))}   # not the identity
    cases = []
    for ci, (N, n, shuffle, seed) in enumerate([(23, 4, False, 0), (40, 3, True, 11), (1, 5, True, 2)]):
        instances = []
        for i in range(N):
            k = rng.randint(1, 4)
            lab = {ents[j]: float(m) for j, m in zip(rng.choice(7, k, replace=False),
                                                     rng.randint(1, 4, k))}
            instances.append(('D%d' % rng.randint(0, 9), tuple(int(t) for t in rng.randint(0, 300, n)), lab))
        given = [(d, list(w), dict(l)) for d, w, l in instances]
        np.random.seed(seed)
        x, y = prepare_mod.instances_and_labels_to_arrays(list(instances), n, class_mapping, np.uint16, shuffle)
        y = y.tocsr()
        y.sort_indices()
        cases.append(dict(window_size=n, shuffle=shuffle, seed=seed, instances=given,
                          class_mapping=class_mapping, x=x.tolist(), x_dtype=str(x.dtype),
                          y_indptr=y.indptr.tolist(), y_indices=y.indices.tolist(),
                          y_data=[float(v) for v in y.data], y_shape=list(y.shape)))
    meta['instances_to_arrays'] = cases
    meta['candidate_centric_label'] = prepare_mod._candidate_centric_label(['E03', 'E01'])

    # ---- a1: _iterate_batches (models.py:351-399) ---------------------------
    cases = []
    for ci, (N, B, seed, shuffle) in enumerate([(1000, 64, 0, True), (1024, 128, 1, True),
                                                (130, 64, 2, False), (63, 64, 3, True),
                                                (4096, 32, 4, True)]):
        mi = models.ModelInterface(B)
        visited = []

        def fn(i, visited=visited):
            visited.append(int(i))
            return np.float32(0.5 + i)
        np.random.seed(seed)
        nb, results = mi._iterate_batches(fn, N, shuffle=shuffle)
        cases.append(dict(N=N, B=B, seed=seed, shuffle=shuffle, num_batches=int(nb),
                          visited=visited, results=[float(r) for r in results]))
    meta['iterate_batches'] = cases

    # non-finite loss -> RuntimeError
    mi = models.ModelInterface(4)
    try:
        mi._iterate_batches(lambda i: np.float32('nan') if i == 1 else np.float32(1), 16)
        meta['iterate_batches_nan_raises'] = False
    except RuntimeError:
        meta['iterate_batches_nan_raises'] = True

    # ---- sparse_to_one_hot_multiple (train.py:186-245) -----------------------
    rng = np.random.RandomState(5)
    rows, cols, vals = [], [], []
    N, Ve, n = 40, 9, 3
    for i in range(N):
        k = rng.randint(1, 4)
        idx = np.sort(rng.choice(Ve, k, replace=False))
        rows += [i] * k
        cols += list(idx)
        vals += [1.0 / k] * k
    y = sp.csr_matrix((np.array(vals, dtype=np.float32), (rows, cols)), shape=(N, Ve))
    x = rng.randint(0, 200, size=(N, n)).astype(np.uint8)
    w = rng.uniform(0.5, 2.0, N).astype(np.float32)
    new_y, (new_x, new_w) = train.sparse_to_one_hot_multiple(y, x, w)
    arrays.update(oh_y_indptr=y.indptr, oh_y_indices=y.indices, oh_y_data=y.data,
                  oh_x=x, oh_w=w, oh_new_y=new_y, oh_new_x=new_x, oh_new_w=new_w)
    meta['one_hot_shape'] = [N, Ve]
    # the documented toy case
    ytoy = sp.csr_matrix(np.array([[0, .5, .5], [1, 0, 0], [0, 0, 1]], dtype=np.float32))
    ty, (tx,) = train.sparse_to_one_hot_multiple(ytoy, np.arange(3, dtype=np.int32)[:, None])
    meta['one_hot_toy'] = dict(y=[int(v) for v in ty], x=[int(v) for v in tx.ravel()])

    meta['error_delta'] = [
        dict(inp=[], out=list(train.error_delta([]))),
        dict(inp=[2.0], out=list(train.error_delta([2.0]))),
        dict(inp=[2.0, 1.5], out=list(train.error_delta([2.0, 1.5]))),
        dict(inp=[2.0, 1.5, 1.8], out=list(train.error_delta([2.0, 1.5, 1.8]))),
    ]

    # ---- train() driver (train.py:262-348): call + dump sequence -------------
    class FakeModel(models.ModelInterface):
        def __init__(self, train_errors):
            models.ModelInterface.__init__(self, 8)
            self.calls = []
            self.train_errors = list(train_errors)
            self.k = 0

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
    drv = []
    for errors, epochs in [([3.0, 2.0, 1.5, 1.2], 3), ([3.0, 2.0, 2.0 + 1e-7, 1.0], 3)]:
        fm = FakeModel(errors)
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, 'model')
            train.train(fm, epochs, out, abort_threshold=1e-5, early_stopping=False,
                        additional_args=[{'args': 1}])
            files = sorted(os.listdir(d))
            counts = {}
            for f in files:
                c = 0
                with open(os.path.join(d, f), 'rb') as fh:
                    while True:
                        try:
                            pickle.load(fh)
                            c += 1
                        except EOFError:
                            break
                counts[f] = c
        drv.append(dict(errors=errors, epochs=epochs, calls=fm.calls, files=files,
                        pickles_per_file=counts))
    meta['train_driver'] = drv

    # ---- inference.WordBatcher (inference.py:28-143) --------------------------
    Vw, B, n, Ve = 300, 4, 3, 7
    table = np.random.RandomState(6).rand(Vw, Ve).astype(np.float32)

    batches = []

    def predict_fn(batch, mask):
        batches.append((batch.copy(), mask.copy()))
        return table[batch.astype(np.int64)]            # (B, n, Ve), depends on ids only

    calls = []

    class CB(object):
        def __call__(self, payload, result, **kw):
            calls.append((list(payload), np.array(result), dict(kw)))

        def should_average_input(self):
            return False
    wb = inference.create(predict_fn, None, B, n, Vw, CB())
    queries = [[5, 9], [1, 2, 3, 4, 5], [7], [11, 12, 13], [20, 21, 22, 23, 24, 25, 26, 27, 28, 29]]
    for qi, q in enumerate(queries):
        wb.submit(list(q), topic_id='t%d' % qi)
    wb.process()
    arrays['wb_table'] = table
    meta['wb'] = dict(Vw=Vw, B=B, n=n, queries=queries, dtype=str(wb.batch.dtype),
                      num_batches=len(batches),
                      call_payloads=[c[0] for c in calls],
                      call_topics=[c[2]['topic_id'] for c in calls])
    for bi, (b, m) in enumerate(batches):
        arrays['wb_batch_%d' % bi] = b
        arrays['wb_mask_%d' % bi] = m
    for ci, c in enumerate(calls):
        arrays['wb_result_%d' % ci] = c[1]
    try:
        wb2 = inference.create(predict_fn, None, 2, 3, Vw, CB())
        wb2.submit(list(range(7)), topic_id='x')
        meta['wb_overlong_raises'] = False
    except RuntimeError:
        meta['wb_overlong_raises'] = True

    # ---- inference.EmbeddingMapper (inference.py:146-167) ---------------------
    Rw = np.random.RandomState(7).randn(50, 6).astype(np.float32)
    seen = []

    class CB2(CB):
        def should_average_input(self):
            return True
    em = inference.create(lambda avg: seen.append(np.array(avg)) or avg[None, :] * 2.0,
                          Rw, 4, 3, 50, CB2())
    calls.clear()
    em.submit([3, 4, 10], topic_id='q')
    arrays['em_Rw'] = Rw
    arrays['em_avg'] = seen[0]
    arrays['em_result'] = calls[0][1]

    # ---- aggregate_distribution (inference.py:170-183) ------------------------
    D = np.random.RandomState(8).dirichlet(np.ones(5), size=4).astype(np.float32)
    D[1, 2] = 0.0
    arrays['agg_in'] = D
    for mode in ['sum', 'product', 'last', 'max', 'identity']:
        arrays['agg_' + mode] = inference.aggregate_distribution(D, mode, 0)
    arrays['agg_zero_case'] = inference.aggregate_distribution(
        np.array([[0, .5], [.5, .5]]), 'product', 0)

    # ---- math_utils.entropy ----------------------------------------------------
    arrays['entropy_in'] = D[0]
    meta['entropy'] = dict(plain=float(math_utils.entropy(D[0])),
                           base2_norm=float(math_utils.entropy(D[0], base=2, normalize=True)))

    # ---- VectorSpaceCallback (query.py:239-370) ---------------------------------
    rng = np.random.RandomState(9)
    Ve, de = 60, 8
    E = rng.randn(Ve, de).astype(np.float32)
    projs = np.tanh(rng.randn(5, de)).astype(np.float32)
    arrays['vs_E'] = E
    arrays['vs_proj'] = projs
    for top in [10, None, 100]:
        ranked = []

        def rank_cb(topic_id, idx, val):
            ranked.append((topic_id, np.array(idx), np.array(val)))
        args = types.SimpleNamespace(top=top)
        margs = types.SimpleNamespace(entity_representation_size=de)
        cb = query.VectorSpaceCallback(E.copy(), args, margs, {i: 'w%d' % i for i in range(20)},
                                       io.StringIO(), rank_cb)
        for qi in range(projs.shape[0]):
            cb([1, 2], projs[qi][None, :].copy(), topic_id='q%d' % qi)
        tag = 'all' if top is None else str(top)
        for qi, (tid, idx, val) in enumerate(ranked):
            arrays['vs_top%s_idx_%d' % (tag, qi)] = idx.astype(np.int64)
            arrays['vs_top%s_val_%d' % (tag, qi)] = val.astype(np.float64)

    # ---- LogLinearCallback (query.py:199-236) -----------------------------------
    rng = np.random.RandomState(10)
    Ve = 25
    ll_inputs = [rng.dirichlet(np.ones(Ve), size=t).astype(np.float32) for t in (1, 3, 6)]
    for qi, P in enumerate(ll_inputs):
        ranked = []

        def rank_cb(topic_id, idx, val):
            ranked.append((np.array(idx), np.array(val)))
        dbg = io.StringIO()
        cb = query.LogLinearCallback(types.SimpleNamespace(), types.SimpleNamespace(),
                                     {i: 'w%d' % i for i in range(10)}, dbg, rank_cb)
        cb(list(range(P.shape[0])), P.copy(), topic_id='q')
        arrays['ll_in_%d' % qi] = P
        arrays['ll_idx_%d' % qi] = ranked[0][0].astype(np.int64)
        arrays['ll_val_%d' % qi] = ranked[0][1].astype(np.float64)
        arrays['ll_entropies_%d' % qi] = np.array(query.compute_normalised_entropy(P, base=2))
    meta['ll_num'] = len(ll_inputs)

    # ---- constants -----------------------------------------------------------------
    meta['constants'] = dict(TRAIN=models.ModelInterface.TRAIN,
                             VALIDATE=models.ModelInterface.VALIDATE,
                             TEST=models.ModelInterface.TEST,
                             OVERFLOW=inference.WordBatcher.OVERFLOW,
                             TRUNCATE=inference.WordBatcher.TRUNCATE)

    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **arrays)
    with open(os.path.join(HERE, 'reference_vectors.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote %d arrays, %d meta entries' % (len(arrays), len(meta)))


if __name__ == '__main__':
    main()
