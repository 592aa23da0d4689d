"""Philox4x32-10 on the CPU + the negative-sample stream of the HIP engine, restated.

TEST INFRASTRUCTURE ONLY (see oracle/sert_oracle.py's header): lets a parity test feed the oracle the SAME negatives the
device draws in its throughput mode, so that mode meets the oracle directly.

The reference draws its negatives with ``RandomStreams(seed).choice(size=(B, z), a=V_e, p=uniform)``
(sert/models.py:947-979): iid uniform over the entities, with replacement, the target not excluded.  Theano's MRG31k3p
stream is not reproducible here (Theano is absent) and is seeded from ``np.random.randint`` anyway (:958-959), so only the
DISTRIBUTION is specified; the engine's stream (sert_amd/csrc/kernels_vs.h: vs_sample_negatives) is

    counter  = (q lo, q hi, pos lo, pos hi)      q = global sample index // 4, sample index = global_row * z + j
    key      = (seed lo, seed hi)
    pos      = 2 * (optimiser updates applied so far)   for training draws, 2 * (evaluation draws so far) + 1 for evaluations
    neg      = (philox4x32_10(counter, key)[index % 4] * V_e) >> 32

Philox4x32-10 itself is Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the
known-answer vectors of its reference implementation (Random123 kat_vectors) pin ``philox4x32_10`` in
tests/test_philox_cpu.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
S32 = np.uint64(32)


def philox4x32_10(ctr, key):
    """ctr (..., 4) uint32, key (2,) or (..., 2) uint32 -> (..., 4) uint32."""
    c = np.asarray(ctr, dtype=np.uint64) & MASK
    k = np.broadcast_to(np.asarray(key, dtype=np.uint64) & MASK, c.shape[:-1] + (2,))
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(10):
        p0 = M0 * c0                     # (< 2^64: both factors < 2^32)
        p1 = M1 * c2
        n0 = (p1 >> S32) ^ c1 ^ k0
        n1 = p1 & MASK
        n2 = (p0 >> S32) ^ c3 ^ k1
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def stream_ids(seed, pos, first, count, num_entities):
    """`count` samples of the stream (seed, pos) starting at global sample index `first` -> int64 (count,)."""
    idx = np.arange(first, first + count, dtype=np.uint64)
    q = idx >> np.uint64(2)
    uq, inv = np.unique(q, return_inverse=True)
    ctr = np.empty((len(uq), 4), dtype=np.uint64)
    ctr[:, 0] = uq & MASK
    ctr[:, 1] = uq >> S32
    ctr[:, 2] = np.uint64(pos) & MASK
    ctr[:, 3] = np.uint64(pos) >> S32
    key = np.array([int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    r = philox4x32_10(ctr, key).astype(np.uint64)
    lane = (idx & np.uint64(3)).astype(np.int64)
    word = r[inv, lane]
    return ((word * np.uint64(num_entities)) >> S32).astype(np.int64)


def training_negatives(seed, updates_applied, batch_size, num_negatives, num_entities, first_row=0, rows=None):
    """The (rows, z) negatives the engine's training step draws when `updates_applied` optimiser updates have been applied
    (= its step counter before the step).  `first_row` / `rows`: a rank's slice of the GLOBAL batch (keyed by global row:
    rank-count invariant)."""
    rows = batch_size if rows is None else rows
    ids = stream_ids(seed, 2 * int(updates_applied), first_row * num_negatives, rows * num_negatives, num_entities)
    return ids.reshape(rows, num_negatives)


def evaluation_negatives(seed, draws_so_far, batch_size, num_negatives, num_entities):
    ids = stream_ids(seed, 2 * int(draws_so_far) + 1, 0, batch_size * num_negatives, num_entities)
    return ids.reshape(batch_size, num_negatives)
