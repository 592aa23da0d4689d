"""Data-parallel host logic (new: the reference is single-device).

One process per GPU on ONE node.  Ranks are started either by ``launch()`` below
(``python bench.py --gpus N`` / ``bin/train.py --gpus N`` spawn their own workers) or by
any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT
(``python -m torch.distributed.run`` works, but nothing here imports PyTorch).

The training pairs of every GLOBAL batch are split evenly over the ranks; each rank
runs forward/backward on its rows; the word-table gradient is reduce-scattered, every
rank applies the dense optimiser to ITS rows of the table and the updated rows are
all-gathered (RCCL over xGMI, called directly from libsert_hip.so on the model's HIP
streams; see csrc/sert_hip.hip).  The small tensors are all-reduced and updated
identically on every rank.

What the host needs beyond that is tiny -- the 128-byte ncclUniqueId, the initial
parameters, barriers, a max over ranks of a wall time -- and goes through a
rendezvous DIRECTORY on the node's shared memory file system (atomic renames, polling):
no sockets, no port beyond the one the launcher already owns, no third-party package.
``host_alltoall`` (the verification transport behind SERT_COMM=host, several ranks on
one GPU) moves its float segments through the same directory.
"""
import atexit
import hashlib
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

_POLL_S = 0.0005
_TIMEOUT_S = float(os.environ.get('SERT_RDZV_TIMEOUT', '600'))


class Context(object):
    def __init__(self, rank=0, local_rank=0, world_size=1):
        self.rank, self.local_rank, self.world_size = rank, local_rank, world_size

    def unique_id(self):
        """A fresh ncclGetUniqueId from rank 0, shipped to everyone.  Collective: every
        rank calls it once per communicator (= per model), in the same order; an id is
        never reused for a second communicator."""
        from sert_amd import _capi
        uid = _capi.comm_unique_id() if self.rank == 0 else None
        return broadcast_object(uid)


class FileStore(object):
    """Key -> bytes over a directory every rank of the node can see.  A value becomes
    visible atomically (write to a private name, then rename); readers poll."""

    def __init__(self, path, rank, world):
        self.path, self.rank, self.world = path, rank, world
        self.seq = 0
        os.makedirs(path, exist_ok=True)

    def _file(self, key):
        return os.path.join(self.path, key)

    def set(self, key, data):
        tmp = self._file('.%s.%d.tmp' % (key, self.rank))
        with open(tmp, 'wb') as f:
            f.write(data)
        os.replace(tmp, self._file(key))

    def get(self, key, timeout=None):
        path = self._file(key)
        deadline = time.time() + (timeout or _TIMEOUT_S)
        delay = _POLL_S
        while True:
            try:
                with open(path, 'rb') as f:
                    return f.read()
            except FileNotFoundError:
                if time.time() > deadline:
                    raise RuntimeError('rendezvous timed out waiting for %r in %s (rank %d of %d)'
                                       % (key, self.path, self.rank, self.world))
                time.sleep(delay)
                delay = min(0.01, delay * 1.5)

    def wait(self, key, timeout=None):
        """Block until `key` is published; returns its path (large values are read by the caller)."""
        path = self._file(key)
        deadline = time.time() + (timeout or _TIMEOUT_S)
        delay = _POLL_S
        while not os.path.exists(path):
            if time.time() > deadline:
                raise RuntimeError('rendezvous timed out waiting for %r in %s (rank %d of %d)'
                                   % (key, self.path, self.rank, self.world))
            time.sleep(delay)
            delay = min(0.01, delay * 1.5)
        return path

    def drop(self, key):
        try:
            os.unlink(self._file(key))
        except OSError:
            pass

    def next_generation(self):
        self.seq += 1
        return self.seq

    def exchange(self, tag, data):
        """Every rank contributes `data` (bytes); returns the list of all contributions in
        rank order.  This rank's file of its PREVIOUS exchange is removed once the current one is
        complete: every rank has then published its current contribution, which it does only after
        it finished reading the previous exchange (broadcasts in between do not matter: the file is
        tracked by name, not by generation arithmetic)."""
        g = self.next_generation()
        mine = '%s.%d.%d' % (tag, g, self.rank)
        self.set(mine, data)
        out = [data if r == self.rank else self.get('%s.%d.%d' % (tag, g, r)) for r in range(self.world)]
        if self._last_exchange_file is not None:
            self.drop(self._last_exchange_file)
        self._last_exchange_file = mine
        return out

    _last_exchange_file = None
    _last_a2a = None


_context = Context()
_store = None


def get_context():
    return _context


def _process_start_time(pid):
    """Field 22 of /proc/<pid>/stat (clock ticks since boot): with the pid, unique per process."""
    try:
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        return ''


def _shared_tmp_base():
    return '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()


def _rendezvous_dir():
    explicit = os.environ.get('SERT_RDZV_DIR')
    if explicit:
        return explicit
    # ranks started by a foreign launcher share its pid as parent and its MASTER_PORT;
    # torch's elastic agent also hands every worker a path inside one per-launch directory
    # ... and that launcher's START TIME: a crashed earlier run with the same port and (in a container,
    # easily) the same parent pid leaves its files behind -- rank 0's cleanup does not run on SIGKILL --
    # and a new run must never see them (stale barrier files, a stale ncclUniqueId)
    parts = [os.environ.get('MASTER_ADDR', ''), os.environ.get('MASTER_PORT', ''), str(os.getppid()),
             _process_start_time(os.getppid())]
    err_file = os.environ.get('TORCHELASTIC_ERROR_FILE')
    if err_file:
        parts.append(os.path.dirname(os.path.dirname(os.path.dirname(err_file))))
    tag = hashlib.sha1('|'.join(parts).encode()).hexdigest()[:16]
    return os.path.join(_shared_tmp_base(), 'sert_rdzv_' + tag)


def init_from_env():
    """Initialise from the launcher's environment.  No-op for WORLD_SIZE <= 1."""
    global _context, _store
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        _context = Context()
        return _context
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # RCCL over dmabuf IPC
    if _store is None:
        _store = FileStore(_rendezvous_dir(), rank, world)
        atexit.register(_cleanup)
    _context = Context(rank, local_rank, world)
    barrier()
    return _context


def set_context(ctx):
    """Tests install a context by hand."""
    global _context
    _context = ctx


def _cleanup():
    global _store
    if _store is not None and _store.rank == 0:
        shutil.rmtree(_store.path, ignore_errors=True)
    _store = None


def shutdown():
    global _context
    if _store is not None:
        st = _store
        barrier()
        # rank 0 removes the directory: only after every other rank has LEFT the barrier
        if st.rank != 0:
            st.set('bye.%d' % st.rank, b'')
        else:
            for r in range(1, st.world):
                st.get('bye.%d' % r)
        _cleanup()
    _context = Context()


def _need_store():
    if _store is None:
        raise RuntimeError('world_size > 1 but distributed.init_from_env() was not called')
    return _store


def broadcast_object(obj, src=0):
    if _context.world_size <= 1:
        return obj
    st = _need_store()
    g = st.next_generation()
    key = 'bc.%d' % g
    if _context.rank == src:
        st.set(key, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        # (kept until shutdown: a broadcast has no completion signal, and they are small or rare)
        return obj
    return pickle.loads(st.get(key))


def broadcast_array(a, src=0):
    """Make a host array identical on every rank (initial parameters)."""
    if _context.world_size <= 1:
        return a
    a = np.ascontiguousarray(a)
    st = _need_store()
    g = st.next_generation()
    key = 'ba.%d' % g
    if _context.rank == src:
        st.set(key, a.tobytes())
        out = a
    else:
        out = np.frombuffer(st.get(key), dtype=a.dtype).reshape(a.shape).copy()
    # the payload can be hundreds of MB: free it as soon as everyone has it
    barrier()
    if _context.rank == src:
        st.drop(key)
    return out


def _all_gather_bytes(tag, data):
    return _need_store().exchange(tag, data)


def host_alltoall(send, send_offsets, send_counts, recv, recv_offsets, recv_counts):
    """The callback of the host-mediated exchange (SERT_COMM=host: several ranks on one GPU, for
    verification; RCCL refuses duplicate devices): rank r's `send` holds send_counts[q] float32 at
    send_offsets[q] for every rank q; on return recv[recv_offsets[q] : + recv_counts[q]] holds what
    rank q addressed to this rank.  Bits are moved, never interpreted.  One file per rank and call
    (the whole send buffer, with its offset table in front); every reader maps the slice addressed to it."""
    world, rank = _context.world_size, _context.rank
    so = np.asarray(send_offsets, dtype=np.int64)
    sc = np.asarray(send_counts, dtype=np.int64)
    ro = np.asarray(recv_offsets, dtype=np.int64)
    rc = np.asarray(recv_counts, dtype=np.int64)
    if world <= 1:
        if rc[0]:
            recv[ro[0]:ro[0] + rc[0]] = send[so[0]:so[0] + sc[0]]
        return
    st = _need_store()
    g = st.next_generation()
    n = int((so + sc).max())
    header = np.concatenate([so, sc]).astype(np.int64)
    st.set('a2a.%d.%d' % (g, rank), header.tobytes() + np.ascontiguousarray(send[:n]).tobytes())
    hbytes = header.nbytes
    for q in range(world):
        if q == rank:
            if rc[q]:
                assert rc[q] == sc[q]
                recv[ro[q]:ro[q] + rc[q]] = send[so[q]:so[q] + sc[q]]
            continue
        path = st.wait('a2a.%d.%d' % (g, q))      # (published atomically: it exists = it is complete)
        with open(path, 'rb') as f:
            hdr = np.frombuffer(f.read(hbytes), dtype=np.int64)
            off, cnt = int(hdr[rank]), int(hdr[world + rank])
            if cnt != int(rc[q]):
                raise RuntimeError('host all-to-all: rank %d sends %d floats to rank %d, which expects %d'
                                   % (q, cnt, rank, int(rc[q])))
            if cnt:
                f.seek(hbytes + 4 * off)
                recv[ro[q]:ro[q] + cnt] = np.frombuffer(f.read(4 * cnt), dtype=np.float32)
    # this rank's file of the previous call can go once everybody has published the current one
    if st._last_a2a is not None:
        st.drop(st._last_a2a)
    st._last_a2a = 'a2a.%d.%d' % (g, rank)


def barrier():
    if _context.world_size > 1:
        _all_gather_bytes('bar', b'')


def all_gather_object(obj):
    """Every rank's (picklable) object, in rank order."""
    if _context.world_size <= 1:
        return [obj]
    return [pickle.loads(b) for b in _all_gather_bytes('ago', pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))]


def all_reduce_max(value):
    if _context.world_size <= 1:
        return float(value)
    parts = _all_gather_bytes('max', np.float64(value).tobytes())
    return float(max(np.frombuffer(p, dtype=np.float64)[0] for p in parts))


def all_reduce_sum_array(a):
    """Sum a host array over ranks (rank order; CPU tests of the data-parallel algebra)."""
    if _context.world_size <= 1:
        return a
    a = np.ascontiguousarray(a)
    parts = _all_gather_bytes('sum', a.tobytes())
    total = np.frombuffer(parts[0], dtype=a.dtype).copy()
    for p in parts[1:]:
        total += np.frombuffer(p, dtype=a.dtype)
    return total.reshape(a.shape)


def shard_rows(num_instances, global_batch, rank, world):
    """Row indices rank `rank` owns: rows [r*B_l, (r+1)*B_l) of every global
    batch; the incomplete tail batch is dropped (sert/models.py:355-359)."""
    local = global_batch // world
    nb = num_instances // global_batch
    base = (np.arange(nb, dtype=np.int64) * global_batch)[:, None]
    offs = np.arange(local, dtype=np.int64)[None, :] + rank * local
    return (base + offs).ravel()


def launch(argv, nproc, env=None, timeout=None):
    """Start `nproc` ranks of `argv` (a command line) on this node, one per GPU, and wait.
    Rank 0 inherits stdout; every rank inherits stderr.  Returns the first non-zero exit
    code (0 if all ranks succeeded); a rank that fails takes the others down."""
    rdzv = tempfile.mkdtemp(prefix='sert_rdzv_', dir=_shared_tmp_base())
    procs = []
    try:
        for rank in range(nproc):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nproc), SERT_RDZV_DIR=rdzv,
                     HSA_ENABLE_IPC_MODE_LEGACY=e.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
            procs.append(subprocess.Popen(argv, env=e, stdout=None if rank == 0 else subprocess.DEVNULL))
        deadline = time.time() + timeout if timeout else None
        rc = 0
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in pending:       # one rank died: the others would wait forever
                        q.terminate()
            if deadline and time.time() > deadline:
                for q in pending:
                    q.kill()
                return 124
            time.sleep(0.02)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdzv, ignore_errors=True)


if __name__ == '__main__':
    # python -m sert_amd.distributed N prog args...   (a minimal one-node launcher)
    sys.exit(launch([sys.executable] + sys.argv[2:], int(sys.argv[1])))
