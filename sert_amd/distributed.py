"""Data-parallel host logic (new: the reference is single-device).

One process per GPU, launched by ``python -m torch.distributed.run`` (or any
launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
The training pairs of every GLOBAL batch are split evenly over the ranks; each
rank runs forward/backward on its rows, the flat gradient buffer (plus the
partial loss sum) is summed over ranks by ONE ``ncclAllReduce`` (RCCL over
xGMI, called directly from libsert_hip.so on the model's HIP stream), and every
rank applies the identical dense optimiser step to its replica.

torch.distributed (gloo, CPU) is used for rendezvous only: shipping the 128-byte
ncclUniqueId, broadcasting initial parameters, barriers and host-side timing
reductions.  No tensor on the data path ever touches PyTorch.
"""
import os

import numpy as np


class Context(object):
    def __init__(self, rank=0, local_rank=0, world_size=1):
        self.rank, self.local_rank, self.world_size = rank, local_rank, world_size

    def unique_id(self):
        """A fresh ncclGetUniqueId from rank 0, shipped to everyone through gloo.
        Collective: every rank calls it once per communicator (= per model), in
        the same order; an id is never reused for a second communicator."""
        from sert_amd import _capi
        uid = _capi.comm_unique_id() if self.rank == 0 else None
        return broadcast_object(uid)


_context = Context()
_pg_ready = False


def get_context():
    return _context


def init_from_env(backend='gloo'):
    """Initialise from the launcher's environment.  No-op for WORLD_SIZE<=1."""
    global _context, _pg_ready
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        _context = Context()
        return _context
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _pg_ready = True
    _context = Context(rank, local_rank, world)
    return _context


def set_context(ctx):
    """Tests install a context by hand."""
    global _context
    _context = ctx


def shutdown():
    global _context, _pg_ready
    if _pg_ready:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    _pg_ready = False
    _context = Context()


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_object(obj, src=0):
    if _context.world_size <= 1:
        return obj
    box = [obj]
    _dist().broadcast_object_list(box, src=src)
    return box[0]


def broadcast_array(a, src=0):
    """Make a host array identical on every rank (initial parameters)."""
    if _context.world_size <= 1:
        return a
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    _dist().broadcast(t, src=src)
    return t.numpy()


def host_allreduce(array):
    """In-place sum of a float32 numpy array over the ranks through gloo -- the callback
    of the host-mediated exchange (SERT_COMM=host: several ranks on one GPU, for
    verification; RCCL refuses duplicate devices)."""
    if _context.world_size <= 1:
        return
    import torch
    t = torch.from_numpy(array)
    _dist().all_reduce(t)


def barrier():
    if _context.world_size > 1:
        _dist().barrier()


def all_reduce_max(value):
    if _context.world_size <= 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum_array(a):
    """Sum a host array over ranks (CPU tests of the data-parallel algebra)."""
    if _context.world_size <= 1:
        return a
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return t.numpy()


def shard_rows(num_instances, global_batch, rank, world):
    """Row indices rank `rank` owns: rows [r*B_l, (r+1)*B_l) of every global
    batch; the incomplete tail batch is dropped (sert/models.py:355-359)."""
    local = global_batch // world
    nb = num_instances // global_batch
    base = (np.arange(nb, dtype=np.int64) * global_batch)[:, None]
    offs = np.arange(local, dtype=np.int64)[None, :] + rank * local
    return (base + offs).ravel()
