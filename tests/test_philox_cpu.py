"""CPU: the oracle's Philox4x32-10 against the known-answer vectors of the algorithm's reference implementation
(Random123, kat_vectors: `philox4x32 10`), and the shape of the negative stream built on it."""
import numpy as np

from oracle import philox as P

KAT = [   # (counter, key, expected) -- Random123 kat_vectors, philox4x32 10 rounds
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        got = P.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(x) for x in got) == want, ([hex(int(x)) for x in got], [hex(x) for x in want])
    # vectorised = one at a time
    ctrs = np.array([k[0] for k in KAT], dtype=np.uint32)
    keys = np.array([k[1] for k in KAT], dtype=np.uint32)
    got = P.philox4x32_10(ctrs, keys)
    assert [tuple(int(x) for x in r) for r in got] == [k[2] for k in KAT]


def test_negative_stream_is_keyed_by_global_row():
    B, z, Ve = 64, 10, 1000
    full = P.training_negatives(1234, 7, B, z, Ve)
    assert full.shape == (B, z) and full.dtype == np.int64
    assert full.min() >= 0 and full.max() < Ve
    # a rank's slice of the global batch sees the same ids (rank-count invariance), also where a Philox counter
    # (4 samples) straddles the slice boundary: z = 10, 13 rows -> sample 130 is lane 2 of its counter
    for first, rows in ((0, 13), (13, 19), (32, 32)):
        part = P.training_negatives(1234, 7, B, z, Ve, first_row=first, rows=rows)
        assert np.array_equal(part, full[first:first + rows])
    assert not np.array_equal(full, P.training_negatives(1234, 8, B, z, Ve))
    assert not np.array_equal(full, P.training_negatives(1235, 7, B, z, Ve))
    assert not np.array_equal(P.evaluation_negatives(1234, 3, B, z, Ve), P.training_negatives(1234, 3, B, z, Ve))


def test_negative_stream_is_uniform():
    ids = P.training_negatives(99, 0, 65536, 10, 1000).ravel()
    counts = np.bincount(ids, minlength=1000)
    # chi-square of 655360 draws over 1000 cells: mean 999, sd ~44.7
    chi = float(((counts - ids.size / 1000.0) ** 2 / (ids.size / 1000.0)).sum())
    assert 999 - 6 * 44.7 < chi < 999 + 6 * 44.7, chi
