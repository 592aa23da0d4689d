"""The inverted index word -> batch rows and its evaluation order, WITHOUT a GPU (sert_debug_word_index_sum): the host
builder of csrc/word_index.h -- what replaces Theano's scatter-add of the embedding-lookup gradient (autodiff of
sert/models.py:180) -- is run on Zipfian batches and walked on the host exactly as the segmented-sum kernels walk it.
Integer-valued source rows make every summation order exact in float32, so the result must EQUAL np.add.at bit for bit:
levels, chunk bounds, partial-row numbering, the row-grouped level 0 with its eight XCD lists (round 4), dense heavy words."""
import numpy as np
import pytest

from sert_amd import _capi as C


def _zipf_ids(rng, nb, B, n, Vw, dtype):
    ranks = np.minimum(rng.zipf(1.1, size=(nb, B, n)) - 1, Vw - 1)
    return rng.permutation(Vw).astype(dtype)[ranks]


@pytest.mark.parametrize('B,n,Vw,d,groups,dense,dtype', [
    (4096, 10, 2000, 8, 1, False, np.uint16),      # three tree levels (a word with > 4096 occurrences), ungrouped
    (4096, 10, 2000, 8, 8, False, np.uint16),      # the same batch cut into 8 row ranges: XCD lists, word-major partial rows
    (5000, 7, 300, 4, 24, False, np.uint16),       # ragged ranges (209 rows, the last one short), three per XCD list
    (1000, 3, 50000, 4, 16, False, np.uint32),     # nearly every word occurs once: level 0 stores finally, no upper level work
    (20000, 5, 100, 4, 8, True, np.uint8),         # dense heavy words (> 4096 occurrences) leave the tree
    (9000, 12, 40, 4, 10, False, np.uint8),        # every word in every range, thousands of chunks: deep tree, 10 ranges
    (640, 2, 200, 4, 1, False, np.uint8),
])
def test_tree_sum_equals_scatter_add(hip_lib, B, n, Vw, d, groups, dense, dtype):
    rng = np.random.RandomState(B + n + Vw + groups)
    ids = _zipf_ids(rng, 2, B, n, Vw, dtype)
    src = rng.randint(-3, 4, size=(B, d)).astype(np.float32)         # small integers: every partial sum is exact
    for batch in (0, 1):
        got, st = C.debug_word_index_sum(ids, Vw, src, batch=batch, row_groups=groups, dense_heavy=dense, divisor=1.0)
        ref = np.zeros((Vw, d), dtype=np.float32)
        np.add.at(ref, ids[batch].astype(np.int64).ravel(), np.repeat(src, n, axis=0))
        assert np.array_equal(got, ref), (batch, st)
        assert st['distinct_words'] == len(np.unique(ids[batch]))
        assert st['row_groups'] == (groups if groups > 1 else 1)
        if dense:
            assert 1 <= st['dense_words'] <= 16
        # every distinct word outside the dense pass gets exactly one final item
        assert st['final_items'] == st['distinct_words'] - st['dense_words']


@pytest.mark.parametrize('B,n,Vw,d,dense,dtype', [
    (4096, 10, 2000, 8, False, np.uint16),      # three levels, Zipf: singletons, mid-size words, multi-chunk words
    (20000, 5, 100, 4, True, np.uint8),         # with the dense heavy words out of the tree
    (1000, 3, 50000, 4, False, np.uint32),      # nearly every item has one entry: the row comes from the descriptor
    (3, 2, 7, 4, False, np.uint8),
])
def test_sorted_level0_equals_scatter_add(hip_lib, B, n, Vw, d, dense, dtype):
    """Level 0 sorted by item length, every item's first row number in its descriptor (word_index.h: sort_level0 -- what the
    vectorspace models upload since round 5): the host walk takes a one-entry item's row from the descriptor as the kernel
    does, checks the order and the row numbers, and must still equal np.add.at; same item and level counts as unsorted."""
    rng = np.random.RandomState(B + n + Vw)
    ids = _zipf_ids(rng, 2, B, n, Vw, dtype)
    src = rng.randint(-3, 4, size=(B, d)).astype(np.float32)
    for batch in (0, 1):
        got, st = C.debug_word_index_sum(ids, Vw, src, batch=batch, dense_heavy=dense, sort_level0=True)
        _, st0 = C.debug_word_index_sum(ids, Vw, src, batch=batch, dense_heavy=dense)
        ref = np.zeros((Vw, d), dtype=np.float32)
        np.add.at(ref, ids[batch].astype(np.int64).ravel(), np.repeat(src, n, axis=0))
        assert np.array_equal(got, ref), (batch, st)
        assert st == st0


def test_row_grouped_level0_item_count_and_single_item_words(hip_lib):
    """Row grouping multiplies the items (one per (range, word, <= 64 occurrences)) and lets words whose occurrences sit in
    one item skip the partial rows: the counts follow from the data."""
    rng = np.random.RandomState(3)
    B, n, Vw = 8192, 10, 20000
    ids = _zipf_ids(rng, 1, B, n, Vw, np.uint16)
    src = np.ones((B, 4), np.float32)
    _, s1 = C.debug_word_index_sum(ids, Vw, src, row_groups=1)
    _, s8 = C.debug_word_index_sum(ids, Vw, src, row_groups=8)
    grp = np.repeat(np.arange(B) // (B // 8), n)
    key = ids[0].astype(np.int64).ravel() * 8 + grp
    _, cnt = np.unique(key, return_counts=True)
    assert s8['level0_items'] == int(((cnt + 63) // 64).sum())
    _, wcnt = np.unique(ids[0], return_counts=True)
    assert s1['level0_items'] == int(((wcnt + 63) // 64).sum())
    assert s8['level0_items'] > s1['level0_items'] and s8['partial_rows'] > s1['partial_rows']


def test_token_id_outside_the_vocabulary_is_refused(hip_lib):
    ids = np.array([[[1, 2], [3, 99]]], dtype=np.uint8)
    with pytest.raises(C.SertError):
        C.debug_word_index_sum(ids, 50, np.ones((2, 4), np.float32))
