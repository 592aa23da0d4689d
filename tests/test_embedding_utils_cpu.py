"""CPU: --representation_initializer (bin/train.py:131-151): word2vec binary round trip and the
initial word table bin/train.py builds from it."""
import importlib.util
import os
import types

import numpy as np

from sert_amd.utils import embedding_utils as EU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train_module():
    spec = importlib.util.spec_from_file_location('bin_train', os.path.join(ROOT, 'bin', 'train.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_word2vec_binary_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    words = ['alpha', 'beta', 'gamma-delta', 'épsilon', 'x']
    vecs = rng.randn(len(words), 7).astype(np.float32)
    path = str(tmp_path / 'vectors.bin')
    EU.save_binary_representations(path, words, vecs)
    back = list(EU.load_binary_representations(path))
    assert [w for w, _ in back] == words
    assert np.array_equal(np.stack([v for _, v in back]), vecs)
    # vocabulary filter: a set, a word -> entry mapping, the id -> word mapping of the meta file
    for vocab in ({'beta', 'x'}, {'beta': 1, 'x': 2}, {0: 'beta', 1: 'x', 2: 'not-in-file'}):
        got = dict(EU.load_binary_representations(path, vocab))
        assert sorted(got) == ['beta', 'x'] and np.array_equal(got['beta'], vecs[1])
    # a file without the optional newline after each vector (the original word2vec tool omits it
    # for the last word) reads the same
    with open(path, 'rb') as f:
        raw = f.read()
    with open(path, 'wb') as f:
        f.write(raw[:-1])
    assert len(list(EU.load_binary_representations(path))) == len(words)


def test_initial_word_table_takes_pretrained_rows(tmp_path):
    train = _train_module()
    rng = np.random.RandomState(1)
    dim = 6
    pre_words = ['apple', 'pear', 'unused']
    pre = rng.randn(3, dim).astype(np.float32)
    path = str(tmp_path / 'pre.bin')
    EU.save_binary_representations(path, pre_words, pre)
    # meta-file shapes: word -> entry(.id), id -> word; 'Apple' is looked up lower-cased
    vocab = ['Apple', 'pear', 'plum', 'fig']
    words = {w: types.SimpleNamespace(id=i) for i, w in enumerate(vocab)}
    tokens = {i: w for i, w in enumerate(vocab)}
    args = types.SimpleNamespace(word_representation_size=dim, representation_initializer=path)
    np.random.seed(5)
    table = train.initial_word_representations(args, words, tokens)
    np.random.seed(5)
    plain = train.initial_word_representations(
        types.SimpleNamespace(word_representation_size=dim, representation_initializer=None), words, tokens)
    assert table.shape == (4, dim) and table.dtype == np.float32
    assert np.array_equal(table[0], pre[0]) and np.array_equal(table[1], pre[1])
    assert np.array_equal(table[2:], plain[2:])          # the other rows keep their Glorot draw
    limit = np.sqrt(6.0 / (4 + dim))
    assert np.abs(plain).max() <= limit
