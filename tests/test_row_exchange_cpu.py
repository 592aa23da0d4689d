"""CPU (no GPU touched): the multi-rank algebra of the word table's exchange BY ROWS
(sert_amd/csrc/kernels_xchg.h), through the library's own list builder (sert_debug_row_lists: the
function sert_upload_dataset runs on the all-gathered touched-row bitmaps), for worlds of 2, 3, 4
and 8 ranks, ragged vocabularies and batches nobody or everybody touches.

Every rank is simulated in this process with numpy: parameters are fetched through the serve / fetch
lists, per-rank gradient rows are returned and added in rank order through the union lists, and the
result must be, bit for bit, what a dense rank-ordered sum over whole tables gives -- and every rank
must see exactly the rows of R_w its batch touches."""
import numpy as np
import pytest

from sert_amd import _capi


def _bitmaps(rng, world, nb, vocab, density):
    bw = (((vocab + 31) // 32) + 3) // 4 * 4
    touched = rng.rand(world, nb, vocab) < density
    touched[:, 0, :] &= False                      # batch 0: nobody touches anything
    if nb > 1:
        touched[:, 1, :] |= True                   # batch 1: everybody touches everything
    bits = np.zeros((world, nb, bw), dtype=np.uint32)
    for r in range(world):
        for b in range(nb):
            w = np.nonzero(touched[r, b])[0]
            np.bitwise_or.at(bits[r, b], w >> 5, (np.uint32(1) << (w & 31).astype(np.uint32)))
    return touched, bits


@pytest.mark.parametrize('world,vocab', [(2, 101), (3, 1000), (4, 257), (8, 5000)])
def test_row_exchange_equals_dense_rank_ordered_sum(hip_lib, world, vocab):
    rng = np.random.RandomState(world * 1000 + vocab)
    nb, d = 4, 8
    R = (-(-vocab // world) + 15) // 16 * 16           # rows per rank, as shard_setup pads them
    touched, bits = _bitmaps(rng, world, nb, vocab, 0.3)
    for b in range(nb):
        lists = [_capi.debug_row_lists(bits, r, R, vocab, b) for r in range(world)]
        owner_lo = [min(vocab, q * R) for q in range(world)]
        owner_hi = [min(vocab, q * R + R) for q in range(world)]
        # ---- consistency of the two views of every transfer ----
        for q in range(world):
            so = np.concatenate([[0], np.cumsum(lists[q]['serve_cnt'])])
            for r in range(world):
                fo = np.concatenate([[0], np.cumsum(lists[r]['fetch_cnt'])])
                served = lists[q]['serve_rows'][so[r]:so[r + 1]]
                fetched = lists[r]['fetch_rows'][fo[q]:fo[q + 1]]
                assert np.array_equal(served, fetched), (b, q, r)       # same rows, same order, both ends
                if q != r:
                    want = np.nonzero(touched[r, b, owner_lo[q]:owner_hi[q]])[0] + owner_lo[q]
                    assert np.array_equal(served, want)
                else:
                    assert len(served) == 0
        # ---- parameters: a rank ends up with the current value of every row it touches ----
        truth = rng.randn(vocab, d).astype(np.float32)
        for r in range(world):
            mine = np.full((vocab, d), np.nan, np.float32)
            mine[owner_lo[r]:owner_hi[r]] = truth[owner_lo[r]:owner_hi[r]]          # owned rows are current
            fo = np.concatenate([[0], np.cumsum(lists[r]['fetch_cnt'])])
            for q in range(world):
                if q == r:
                    continue
                so = np.concatenate([[0], np.cumsum(lists[q]['serve_cnt'])])
                packet = truth[lists[q]['serve_rows'][so[r]:so[r + 1]]]            # what q packs for r
                mine[lists[r]['fetch_rows'][fo[q]:fo[q + 1]]] = packet              # where r unpacks it
            need = np.nonzero(touched[r, b])[0]
            assert np.array_equal(mine[need], truth[need])
        # ---- gradients: the owner's rank-ordered sum of the rows returned to it ----
        grads = [np.where(touched[r, b][:, None], rng.randn(vocab, d), 0).astype(np.float32) for r in range(world)]
        dense = np.zeros((vocab, d), np.float32)
        for r in range(world):                                                  # rank order, as the kernel adds
            dense = np.where(touched[r, b][:, None], dense + grads[r], dense)
        for q in range(world):
            L = lists[q]
            so = np.concatenate([[0], np.cumsum(L['serve_cnt'])])
            recv = np.zeros((max(1, int(so[-1])), d), np.float32)
            for r in range(world):
                if r == q:
                    continue
                fo = np.concatenate([[0], np.cumsum(lists[r]['fetch_cnt'])])
                recv[so[r]:so[r + 1]] = grads[r][lists[r]['fetch_rows'][fo[q]:fo[q + 1]]]   # r packs, q receives
            got = np.zeros((vocab, d), np.float32)
            for u, w in enumerate(L['union_rows']):
                acc = np.zeros(d, np.float32)
                for e in L['ent'][L['ptr'][u]:L['ptr'][u + 1]]:
                    acc = acc + (grads[q][w] if e < 0 else recv[e])
                got[w] = acc
            lo, hi = owner_lo[q], owner_hi[q]
            assert np.array_equal(got[lo:hi], dense[lo:hi]), (b, q)
            any_touch = touched[:, b, lo:hi].any(axis=0)
            assert np.array_equal(L['union_rows'], np.nonzero(any_touch)[0] + lo)
        assert lists[0]['max_xfer_rows'] >= max(len(lists[0]['serve_rows']), len(lists[0]['fetch_rows']))


def test_bytes_moved_against_zero1(hip_lib):
    """Counted bytes of the two exchanges at a C2-like touch pattern (Zipf(1.1) tokens, window 10): by
    rows a rank moves what its batch touches; ZeRO-1 moves the table twice whatever the batch."""
    rng = np.random.RandomState(0)
    world, vocab, B, n, d = 8, 20000, 8192, 10, 128
    bw = (((vocab + 31) // 32) + 3) // 4 * 4
    bits = np.zeros((world, 1, bw), dtype=np.uint32)
    perm = rng.permutation(vocab)
    for r in range(world):
        w = np.unique(perm[np.minimum(rng.zipf(1.1, size=B * n) - 1, vocab - 1)])
        np.bitwise_or.at(bits[r, 0], w >> 5, (np.uint32(1) << (w & 31).astype(np.uint32)))
    R = (-(-vocab // world) + 15) // 16 * 16
    L = _capi.debug_row_lists(bits, 0, R, vocab, 0)
    rows_bytes = 2 * (len(L['serve_rows']) + len(L['fetch_rows'])) * d * 4          # params + gradients, out + in
    zero1_bytes = 2 * 2 * (world - 1) / world * (R * world) * d * 4
    assert rows_bytes < 0.75 * zero1_bytes, (rows_bytes, zero1_bytes)
