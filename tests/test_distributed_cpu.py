"""CPU, world_size 2: the data-parallel algebra the HIP engine relies on, over the product's
own rendezvous (sert_amd.distributed: a shared-memory directory, no PyTorch) and, as a
cross-check of that transport, over torch.distributed's gloo backend (tests only).

Each rank owns rows [r*B_l, (r+1)*B_l) of every GLOBAL batch
(distributed.shard_rows), computes the data-term gradient of its rows with the
GLOBAL 1/B scaling, the per-rank gradients are summed (all-reduce), and the L2
term + optimiser step are applied once, identically, on every rank.  Checked
here with the oracle as the per-rank compute: the result must equal the
single-process full-batch step.
"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from sert_amd import distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_partition():
    N, B, world = 1000, 64, 4
    parts = [distributed.shard_rows(N, B, r, world) for r in range(world)]
    allrows = np.sort(np.concatenate(parts))
    assert np.array_equal(allrows, np.arange((N // B) * B))          # tail dropped
    for r, p in enumerate(parts):
        assert len(p) == (N // B) * (B // world)
        # local batch j of rank r = rows [j*B + r*B_l, j*B + (r+1)*B_l)
        assert np.array_equal(p[:B // world], np.arange(r * 16, (r + 1) * 16))
        assert np.array_equal(p[16:32], 64 + np.arange(r * 16, (r + 1) * 16))


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from oracle import sert_oracle as O
    from sert_amd import distributed as D

    ctx = D.init_from_env()
    assert ctx.world_size == 2
    if sys.argv[1] == 'gloo':
        # same algebra, sums carried by gloo instead of the product's store
        import torch, torch.distributed as dist
        dist.init_process_group('gloo', rank=ctx.rank, world_size=ctx.world_size)
        def _gloo_sum(a):
            t = torch.from_numpy(np.ascontiguousarray(a).copy())
            dist.all_reduce(t)
            return t.numpy()
        D.all_reduce_sum_array = _gloo_sum
    rng = np.random.RandomState(0)            # same data on every rank
    B, n, z, Vw, Ve, dw, de = 16, 3, 4, 40, 9, 6, 5
    lam = 0.01
    Rw, Re = O.glorot_uniform(rng, (Vw, dw)), O.glorot_uniform(rng, (Ve, de))
    W, b = O.glorot_uniform(rng, (dw, de)), (0.1 * rng.randn(de)).astype(np.float32)
    X = rng.randint(0, Vw, (2 * B, n)); y = rng.randint(0, Ve, 2 * B)
    w = rng.uniform(.5, 2, 2 * B).astype(np.float32)
    neg = rng.randint(0, Ve, (2 * B, z))

    # reference: one process, full global batches
    full = O.VectorSpaceOracle(B, n, z, Rw, Re, W, b, lam)
    ref_losses = [full.train_step(X[j*B:(j+1)*B], y[j*B:(j+1)*B], w[j*B:(j+1)*B], neg[j*B:(j+1)*B])
                  for j in range(2)]

    # data parallel: this rank's rows, global 1/B scaling, summed gradients
    rows = D.shard_rows(2 * B, B, ctx.rank, ctx.world_size)
    Bl = B // ctx.world_size
    dp = O.VectorSpaceOracle(B, n, z, Rw, Re, W, b, lam)   # .B = GLOBAL batch: scales and L2
    losses = []
    for j in range(2):
        r = rows[j*Bl:(j+1)*Bl]
        f = dp.forward(X[r], y[r], neg[r])
        # data term of the loss and of the gradients for the local rows only
        lam_saved, dp.lam = dp.lam, 0.0
        _, grads, _ = dp.loss_and_grads(X[r], y[r], w[r], neg[r])
        dp.lam = lam_saved
        # loss_and_grads divides by len(local rows); rescale to the GLOBAL batch
        scale = np.float32(len(r)) / np.float32(B)
        grads = [g * scale for g in grads]
        local_loss_sum = np.array([np.sum(f['loss'] * w[r], dtype=np.float64)])
        flat = np.concatenate([g.ravel() for g in grads] + [local_loss_sum.astype(np.float32)])
        flat = D.all_reduce_sum_array(flat)                  # ONE exchange per step
        out, o = [], 0
        for g in grads:
            out.append(flat[o:o + g.size].reshape(g.shape)); o += g.size
        loss = np.float32(flat[o] / B) + dp.regularizer()
        k = np.float32(lam) / np.float32(B)                 # L2 once, after the reduce
        out[0] = out[0] + k * dp.R_e; out[1] = out[1] + k * dp.R_w; out[2] = out[2] + k * dp.W
        dp.opt.update(dp.params(), out)
        losses.append(loss)
    for a, c in zip(losses, ref_losses):
        assert abs(a - c) <= 2e-6 * abs(c), (a, c)
    for p, q in zip(dp.params(), full.params()):
        assert np.abs(p - q).max() <= 1e-5 * np.abs(q).max()
    # replicas stay identical
    chk = D.all_reduce_sum_array(np.concatenate([p.ravel() for p in dp.params()]).astype(np.float64))
    mine = np.concatenate([p.ravel() for p in dp.params()]).astype(np.float64)
    assert np.allclose(chk, 2 * mine, rtol=0, atol=0)
    # object / array broadcast used for the ncclUniqueId and the initial parameters
    assert D.broadcast_object('id-from-%%d' %% ctx.rank) == 'id-from-0'
    a = D.broadcast_array(np.full(3, ctx.rank, dtype=np.float32))
    assert np.array_equal(a, np.zeros(3, np.float32))
    assert D.all_reduce_max(float(ctx.rank)) == 1.0
    D.barrier()
    D.shutdown()
    print('rank %%d ok' %% ctx.rank)
''')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('transport', ['store', 'gloo'])
def test_two_ranks_equal_single_process(tmp_path, transport):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), transport], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d ok' % rank in out


ROWS_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from sert_amd import _capi, distributed as D

    ctx = D.init_from_env()
    W, me = ctx.world_size, ctx.rank
    rng = np.random.RandomState(123)                 # same bitmaps and data on every rank
    vocab, d, nb = 1000, 8, 3
    R = (-(-vocab // W) + 15) // 16 * 16
    bw = (((vocab + 31) // 32) + 3) // 4 * 4
    touched = rng.rand(W, nb, vocab) < 0.35
    bits = np.zeros((W, nb, bw), dtype=np.uint32)
    for r in range(W):
        for b in range(nb):
            w = np.nonzero(touched[r, b])[0]
            np.bitwise_or.at(bits[r, b], w >> 5, (np.uint32(1) << (w & 31).astype(np.uint32)))
    lo, hi = min(vocab, me * R), min(vocab, me * R + R)
    table = rng.randn(vocab, d).astype(np.float32)   # the "true" parameters; this rank holds its rows only
    for b in range(nb):
        L = _capi.debug_row_lists(bits, me, R, vocab, b)
        scnt, fcnt = L['serve_cnt'].astype(np.int64) * d, L['fetch_cnt'].astype(np.int64) * d
        soff = np.concatenate([[0], np.cumsum(scnt)[:-1]]); foff = np.concatenate([[0], np.cumsum(fcnt)[:-1]])
        # ---- phase P: owners pack and send, this rank unpacks ----
        mine = np.full((vocab, d), np.nan, np.float32)
        mine[lo:hi] = table[lo:hi]
        send = np.ascontiguousarray(mine[L['serve_rows']].ravel()) if len(L['serve_rows']) else np.zeros(1, np.float32)
        recv = np.zeros(max(1, int(fcnt.sum())), np.float32)
        D.host_alltoall(send, soff, scnt, recv, foff, fcnt)
        if len(L['fetch_rows']):
            mine[L['fetch_rows']] = recv[:fcnt.sum()].reshape(-1, d)
        need = np.nonzero(touched[me, b])[0]
        assert np.array_equal(mine[need], table[need]), ('params', me, b)
        # ---- phase G: gradient rows go back, rank-ordered sum at the owner ----
        grads = [np.where(touched[r, b][:, None], np.random.RandomState(1000 * b + r).randn(vocab, d), 0).astype(np.float32)
                 for r in range(W)]
        g = grads[me].copy()
        send = np.ascontiguousarray(g[L['fetch_rows']].ravel()) if len(L['fetch_rows']) else np.zeros(1, np.float32)
        recv = np.zeros(max(1, int(scnt.sum())), np.float32)
        D.host_alltoall(send, foff, fcnt, recv, soff, scnt)
        rows = recv[:scnt.sum()].reshape(-1, d)
        for u, wrow in enumerate(L['union_rows']):
            acc = np.zeros(d, np.float32)
            for e in L['ent'][L['ptr'][u]:L['ptr'][u + 1]]:
                acc = acc + (g[wrow] if e < 0 else rows[e])
            g[wrow] = acc
        dense = np.zeros((vocab, d), np.float32)
        for r in range(W):
            dense = np.where(touched[r, b][:, None], dense + grads[r], dense)
        assert np.array_equal(g[lo:hi], dense[lo:hi]), ('grads', me, b)
    D.barrier()
    D.shutdown()
    print('rank %%d ok' %% me)
''')


@pytest.mark.parametrize('world', [2, 3])
def test_row_exchange_over_the_store_transport(tmp_path, world):
    """World 2 and 3, one process per rank, CPU only: the row exchange of the data-parallel word table
    (the lists of sert_debug_row_lists = the upload's own builder) carried by the product's host
    transport, sert_amd.distributed.host_alltoall -- every rank ends up with the current value of
    exactly the rows its batch touches, and every owner with the rank-ordered sum of the gradient rows."""
    script = tmp_path / 'rows_worker.py'
    script.write_text(ROWS_WORKER % {'root': ROOT})
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        assert p.returncode == 0, out.decode()
        assert 'rank %d ok' % rank in out.decode()


RDZV_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
from sert_amd import distributed as dist
t0 = time.time()
try:
    ctx = dist.init_from_env()            # FileStore + the first barrier: every rank of the world must arrive
    vals = dist.all_gather_object(('rank', ctx.rank))
    assert [v[1] for v in vals] == list(range(ctx.world_size)), vals
    assert dist.all_reduce_max(float(ctx.rank)) == ctx.world_size - 1
    dist.barrier()
    dist.shutdown()
    print('OK %%d %%.2f' %% (ctx.rank, time.time() - t0))
except RuntimeError as e:
    print('TIMEOUT %%s %%.2f %%s' %% (os.environ['RANK'], time.time() - t0, str(e)[:400].replace(chr(10), ' ')))
    sys.exit(3)
'''


def _spawn_world(tmp_path, world, absent=(), timeout_s='3'):
    import subprocess
    import sys
    rd = str(tmp_path / 'rdzv')
    procs = []
    for r in range(world):
        if r in absent:
            continue
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29533', SERT_RDZV_DIR=rd, SERT_RDZV_TIMEOUT=timeout_s)
        procs.append((r, subprocess.Popen([sys.executable, '-c', RDZV_WORKER % dict(root=ROOT)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    out = {}
    for r, p in procs:
        so, se = p.communicate(timeout=120)
        out[r] = (p.returncode, so.decode().strip(), se.decode()[-300:])
    return out


def test_world_of_eight_rendezvous_and_host_collectives(tmp_path):
    """First contact of an 8-rank launch with the rendezvous must be boring: eight processes (no GPU, no RCCL) meet over the
    FileStore, run the host-side collectives bench.py and the models use around the data path, and shut down."""
    out = _spawn_world(tmp_path, 8, timeout_s='60')
    assert sorted(out) == list(range(8))
    for r, (rc, so, se) in out.items():
        assert rc == 0 and so.startswith('OK %d' % r), (r, rc, so, se)


def test_world_of_eight_times_out_cleanly_when_a_rank_never_arrives(tmp_path):
    """... and when one of the eight never starts, the other seven do not hang: each raises the rendezvous error naming the
    key, the directory, its rank and the world within SERT_RDZV_TIMEOUT."""
    out = _spawn_world(tmp_path, 8, absent=(5,), timeout_s='3')
    assert sorted(out) == [0, 1, 2, 3, 4, 6, 7]
    for r, (rc, so, se) in out.items():
        assert rc == 3 and so.startswith('TIMEOUT %d' % r), (r, rc, so, se)
        assert 'rendezvous timed out' in so and 'of 8' in so, so
        assert float(so.split()[2]) < 30.0, so
