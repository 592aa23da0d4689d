"""CPU: the instance generator (bin/prepare.py equivalent) and the TREC helpers."""
import io
import os
import pickle
import subprocess
import sys

import numpy as np
import scipy.sparse as sp

from sert_amd import prepare as prep
from sert_amd.utils import trec_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_windows_stride_padding():
    ids = list(range(1, 8))                                  # 7 tokens
    assert prep.windows(ids, 3, 3, padding_id=0) == [(1, 2, 3), (4, 5, 6), (7, 0, 0)]
    assert prep.windows(ids, 3, 3, padding_id=None) == [(1, 2, 3), (4, 5, 6)]
    assert prep.windows(ids, 3, 1, padding_id=None) == [(1, 2, 3), (2, 3, 4), (3, 4, 5), (4, 5, 6), (5, 6, 7)]
    assert prep.windows([1, 2], 4, 4, padding_id=0) == [(1, 2, 0, 0)]
    assert prep.windows([], 4, 4, padding_id=0) == []
    assert prep.windows([1, 2], 4, 4, padding_id=None) == []


def test_tokenize():
    assert prep.tokenize('Hello, World-2016! 42 x9') == ['hello', 'world', '<num>', '<num>', 'x9']


def _corpus(tmp_path, ndocs=12):
    rng = np.random.RandomState(0)
    topics = ['alpha beta gamma delta', 'kappa lambda sigma omega', 'red green blue yellow']
    docs, assocs = [], []
    for d in range(ndocs):
        t = d % 3
        wordsl = rng.choice(topics[t].split(), size=17 + d).tolist() + ['the', 'x', str(1000 + d)]
        docs.append('<DOC>\n<DOCNO> D%03d </DOCNO>\n<TEXT>\n%s\n</TEXT>\n</DOC>\n' % (d, ' '.join(wordsl)))
        assocs.append('E%d D%03d 1' % (t, d))
        if d % 4 == 0:
            assocs.append('E%d D%03d 1' % ((t + 1) % 3, d))
    docs.append('<DOC>\n<DOCNO> ORPHAN </DOCNO>\n<TEXT> alpha beta </TEXT>\n</DOC>\n')
    (tmp_path / 'docs.trectext').write_text(''.join(docs))
    (tmp_path / 'assocs').write_text('\n'.join(assocs) + '\nE9 MISSING 1\n')
    return tmp_path


def _run(tmp_path, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, 'bin', 'prepare.py'), '--seed', '3',
           str(tmp_path / 'docs.trectext'), '--assoc_path', str(tmp_path / 'assocs'),
           '--window_size', '4', '--vocabulary_min_count', '1', '--validation_set_ratio', '0.2',
           '--meta_output', str(tmp_path / 'meta'), '--data_output', str(tmp_path / 'data.npz'),
           '--loglevel', 'ERROR'] + list(extra)
    subprocess.check_call(cmd)
    data = np.load(str(tmp_path / 'data.npz'), allow_pickle=True)
    with open(str(tmp_path / 'meta'), 'rb') as f:
        meta = [pickle.load(f) for _ in range(5)]
    return data, meta


def test_prepare_outputs_and_formats(tmp_path):
    _corpus(tmp_path)
    data, (pargs, words, tokens, ent_inv, docs_per_entity) = _run(tmp_path)
    x, y, w = data['x_train'], data['y_train'][()], data['w_train']
    xv, yv = data['x_validate'], data['y_validate'][()]
    assert pargs.window_size == 4 and pargs.stride == 4
    assert x.dtype == np.min_scalar_type(len(words) - 1) and x.shape[1] == 4
    assert sp.isspmatrix_csr(y) and y.dtype == np.float32 and y.shape == (x.shape[0], len(ent_inv))
    assert np.allclose(np.asarray(y.sum(axis=1)).ravel(), 1.0)          # label distributions
    assert set(np.unique(np.diff(y.indptr))) <= {1, 2}                  # 1 or 2 entities per doc
    assert w.dtype == np.float32 and w.shape == (x.shape[0],) and w.min() >= 1.0
    assert len(ent_inv) == 3 and set(ent_inv.values()) == {'E0', 'E1', 'E2'}
    assert 'E9' not in docs_per_entity                                  # association to a missing doc
    assert words['</s>'].id == 0 and tokens[0] == '</s>' and 'the' not in words and 'x' not in words
    assert '<num>' in words
    n_total = x.shape[0] + xv.shape[0]
    assert abs(xv.shape[0] / float(n_total) - 0.2) < 0.05
    # every window holds in-vocabulary ids; padding only at the end of a window
    for row in np.concatenate([x, xv]):
        assert row.max() < len(words)
        nz = np.nonzero(row == 0)[0]
        assert len(nz) == 0 or np.array_equal(nz, np.arange(nz[0], 4))
    # deterministic under the seed
    os.remove(str(tmp_path / 'meta'))
    os.remove(str(tmp_path / 'data.npz'))
    data2, _ = _run(tmp_path)
    assert np.array_equal(data2['x_train'], x) and np.array_equal(data2['w_train'], w)


def test_prepare_resample_overlapping_no_weights(tmp_path):
    _corpus(tmp_path)
    data, (pargs, words, _, ent_inv, _) = _run(
        tmp_path, ['--overlapping', '--resample', '--no_instance_weights', '--no_padding'])
    assert pargs.stride == 1 and 'w_train' not in data.files and '</s>' not in words
    y = data['y_train'][()]
    yv = data['y_validate'][()]
    # resampling equalises the number of instances per distinct label set
    keys = collections_counter_rows(sp.vstack([y, yv]).tocsr())
    assert len(set(keys.values())) == 1


def collections_counter_rows(y):
    import collections
    c = collections.Counter()
    for i in range(y.shape[0]):
        c[tuple(y.indices[y.indptr[i]:y.indptr[i + 1]])] += 1
    return c


def test_trec_utils_roundtrip_and_metrics():
    topics = trec_utils.parse_topics(io.StringIO(u'0;bathroom mats squeegees\n7;Dining   Gadgets-2\n'))
    assert list(topics.items()) == [('0', 'bathroom mats squeegees'), ('7', 'Dining   Gadgets-2')]
    assert trec_utils.parse_query(topics['7']) == ['dining', 'gadgets-2']
    out = io.StringIO()
    trec_utils.write_run('m', {'0': [(0.2, 'B'), (0.9, 'A'), (0.2, 'C')]}, out)
    lines = out.getvalue().splitlines()
    assert lines[0].split()[:4] == ['0', 'Q0', 'A', '1'] and lines[1].split()[2] == 'C'   # ties: id descending, as trec_eval re-sorts
    run = trec_utils.parse_run(io.StringIO(out.getvalue()))
    qrels = trec_utils.parse_qrels(io.StringIO(u'0 0 A 1.0\n0 0 C 1.0\n'))
    res = trec_utils.evaluate_run(run, qrels, k=100)
    # ranking A, C, B (ties broken by doc id descending, as trec_eval): both relevant on top
    assert abs(res['ndcg_cut_100'] - 1.0) < 1e-12 and abs(res['map'] - 1.0) < 1e-12
    qrels2 = trec_utils.parse_qrels(io.StringIO(u'0 0 B 1.0\n'))
    res2 = trec_utils.evaluate_run(run, qrels2, k=100)
    assert abs(res2['ndcg_cut_100'] - 1.0 / np.log2(4)) < 1e-12 and abs(res2['map'] - 1.0 / 3) < 1e-12
