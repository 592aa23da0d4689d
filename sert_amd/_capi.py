"""ctypes binding of include/sert_hip.h (libsert_hip.so).

There is deliberately no fallback: if the HIP library is missing or a call
fails, an exception is raised.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SERT_LIB: load another build of the library (A/B experiments with tools/build_variant.sh)
LIB_PATH = os.environ.get('SERT_LIB') or os.path.join(_HERE, 'libsert_hip.so')

KIND_LOGLINEAR, KIND_VECTORSPACE, KIND_VECTORSPACE_SOFTMAX = 0, 1, 2
SPLIT_TRAIN, SPLIT_VALIDATE = 0, 1

T_RW, T_RE, T_W, T_B = 0, 1, 2, 3
T_STATE0_RW, T_STATE0_RE, T_STATE0_W, T_STATE0_B = 4, 5, 6, 7
T_STATE1_RW, T_STATE1_RE, T_STATE1_W, T_STATE1_B = 8, 9, 10, 11
T_GRAD_RW, T_GRAD_RE, T_GRAD_W, T_GRAD_B = 12, 13, 14, 15
T_ACT_H, T_ACT_T, T_ACT_DA, T_ACT_DH, T_ACT_ROWLOSS = 16, 17, 18, 19, 20

COMM_ID_BYTES = 128

# every symbol include/sert_hip.h declares
EXPORTS = [
    'sert_create', 'sert_destroy', 'sert_last_error', 'sert_device_info', 'sert_device_count',
    'sert_set_tensor', 'sert_get_tensor', 'sert_tensor_size', 'sert_set_step', 'sert_get_step',
    'sert_set_eval_draws', 'sert_get_eval_draws', 'sert_negatives_of_step',
    'sert_upload_dataset', 'sert_train_batch', 'sert_hint_next_batch', 'sert_train_batches',
    'sert_eval_batch', 'sert_eval_batches',
    'sert_predict_project', 'sert_predict_tokens', 'sert_score_topk',
    'sert_scorer_create', 'sert_scorer_destroy', 'sert_scorer_topk', 'sert_scorer_scores',
    'sert_host_alloc', 'sert_host_free',
    'sert_comm_unique_id', 'sert_comm_init', 'sert_comm_init_host', 'sert_comm_destroy', 'sert_comm_stats',
    'sert_synchronize', 'sert_timing_enable', 'sert_timing_reset', 'sert_timing_count',
    'sert_timing_name', 'sert_timing_avg_us', 'sert_timing_launches', 'sert_bench_gemm', 'sert_debug_gemm', 'sert_debug_gemm_splitk', 'sert_debug_gemm_longk', 'sert_bench_memory', 'sert_debug_row_lists', 'sert_debug_word_index_sum',
    'sert_debug_update_counts', 'sert_debug_poison_scratch',
    'sert_profile_range_push', 'sert_profile_range_pop',
]


# int (*sert_alltoall_fn)(void* user, const float* send, const int64_t* send_offsets, const int64_t* send_counts,
#                         float* recv, const int64_t* recv_offsets, const int64_t* recv_counts)
_F32P, _I64P = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
ALLTOALL_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _F32P, _I64P, _I64P, _F32P, _I64P, _I64P)


class SertConfig(ctypes.Structure):
    _fields_ = [
        ('struct_size', ctypes.c_uint32),
        ('kind', ctypes.c_int32),
        ('batch_size', ctypes.c_int32),
        ('global_batch_size', ctypes.c_int32),
        ('window_size', ctypes.c_int32),
        ('vocab_size', ctypes.c_int32),
        ('num_entities', ctypes.c_int32),
        ('word_dim', ctypes.c_int32),
        ('entity_dim', ctypes.c_int32),
        ('num_negatives', ctypes.c_int32),
        ('id_bytes', ctypes.c_int32),
        ('device', ctypes.c_int32),
        ('keep_grads', ctypes.c_int32),
        ('deterministic', ctypes.c_int32),
        ('inference_only', ctypes.c_int32),
        ('lambda_', ctypes.c_float),
        ('lr', ctypes.c_float),
        ('beta1', ctypes.c_float),
        ('beta2', ctypes.c_float),
        ('eps', ctypes.c_float),
        ('seed', ctypes.c_uint64),
    ]


class SertError(RuntimeError):
    pass


_lib = None


def load():
    """Load libsert_hip.so (once) and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SertError(
            'libsert_hip.so not found at %s; build it with '
            '`python -m sert_amd._build` (hipcc --offload-arch=gfx950). '
            'There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    fp = ctypes.c_void_p  # host float*/int* passed as raw addresses
    lib.sert_create.argtypes = [ctypes.POINTER(SertConfig), ctypes.POINTER(vp)]
    lib.sert_destroy.argtypes = [vp]
    lib.sert_last_error.restype = ctypes.c_char_p
    lib.sert_device_info.argtypes = [ctypes.c_int, ctypes.c_char_p, sz]
    lib.sert_device_count.argtypes = []
    lib.sert_set_tensor.argtypes = [vp, ctypes.c_int, fp, sz]
    lib.sert_get_tensor.argtypes = [vp, ctypes.c_int, fp, sz]
    lib.sert_tensor_size.argtypes = [vp, ctypes.c_int]
    lib.sert_tensor_size.restype = sz
    lib.sert_set_step.argtypes = [vp, i64]
    lib.sert_get_step.argtypes = [vp]
    lib.sert_get_step.restype = i64
    lib.sert_set_eval_draws.argtypes = [vp, i64]
    lib.sert_get_eval_draws.argtypes = [vp]
    lib.sert_get_eval_draws.restype = i64
    lib.sert_negatives_of_step.argtypes = [vp, i64, ctypes.c_int, vp]
    lib.sert_upload_dataset.argtypes = [vp, ctypes.c_int, fp, fp, fp, fp, fp, fp, i64]
    lib.sert_train_batch.argtypes = [vp, i64, fp, ctypes.POINTER(ctypes.c_float)]
    lib.sert_train_batches.argtypes = [vp, fp, i64, fp]
    lib.sert_hint_next_batch.argtypes = [vp, i64]
    lib.sert_eval_batch.argtypes = [vp, ctypes.c_int, i64, fp, ctypes.POINTER(ctypes.c_float)]
    lib.sert_eval_batches.argtypes = [vp, ctypes.c_int, ctypes.c_void_p, i64, ctypes.c_void_p]
    lib.sert_predict_project.argtypes = [vp, fp, i64, fp]
    lib.sert_predict_tokens.argtypes = [vp, fp, i64, fp]
    lib.sert_score_topk.argtypes = [ctypes.c_int, fp, i64, i32, fp, i64, i32, fp, fp]
    lib.sert_scorer_create.argtypes = [ctypes.c_int, fp, i64, i32, ctypes.POINTER(vp)]
    lib.sert_scorer_destroy.argtypes = [vp]
    lib.sert_scorer_topk.argtypes = [vp, fp, i64, i32, fp, fp]
    lib.sert_scorer_scores.argtypes = [vp, fp, i64, fp]
    lib.sert_host_alloc.argtypes = [ctypes.POINTER(vp), sz]
    lib.sert_host_free.argtypes = [vp]
    lib.sert_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.sert_comm_init.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.sert_comm_destroy.argtypes = [vp]
    lib.sert_comm_init_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, ALLTOALL_FN, vp]
    lib.sert_comm_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.sert_debug_row_lists.argtypes = [fp, ctypes.c_int, ctypes.c_int, i64, i64, i64, i64, i64] + [fp] * 7 + [i64, fp]
    lib.sert_synchronize.argtypes = [vp]
    lib.sert_profile_range_push.argtypes = [ctypes.c_char_p]
    lib.sert_profile_range_pop.argtypes = []
    lib.sert_timing_enable.argtypes = [vp, ctypes.c_int]
    lib.sert_timing_reset.argtypes = [vp]
    lib.sert_timing_count.argtypes = [vp]
    lib.sert_timing_name.argtypes = [vp, ctypes.c_int]
    lib.sert_timing_name.restype = ctypes.c_char_p
    lib.sert_timing_avg_us.argtypes = [vp, ctypes.c_int]
    lib.sert_timing_avg_us.restype = ctypes.c_double
    lib.sert_timing_launches.argtypes = [vp, ctypes.c_int]
    lib.sert_timing_launches.restype = ctypes.c_double
    lib.sert_bench_gemm.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double)]
    lib.sert_bench_memory.argtypes = [ctypes.c_int, ctypes.c_int, sz, sz, ctypes.c_int, ctypes.c_int, sz, ctypes.c_int,
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise SertError(load().sert_last_error().decode('utf-8', 'replace'))


def _addr(a):
    return None if a is None else a.ctypes.data


def device_count():
    return load().sert_device_count()


def device_info(device=0):
    buf = ctypes.create_string_buffer(512)
    n = load().sert_device_info(device, buf, 512)
    if n < 0:
        raise SertError(load().sert_last_error().decode())
    return buf.value.decode()


def require_gpu():
    """Raise unless a HIP device is visible (the product has no CPU path)."""
    n = device_count()
    if n is None or n <= 0:
        raise SertError('no HIP device visible: sert_amd runs on MI355X (gfx950) only; '
                        'there is no CPU fallback.')
    return n


class Engine(object):
    """Thin owner of one sert_model handle."""

    def __init__(self, **cfg):
        lib = load()
        c = SertConfig()
        c.struct_size = ctypes.sizeof(SertConfig)
        for k, v in cfg.items():
            if not hasattr(c, k):
                raise TypeError('unknown sert_config field %r' % k)
            setattr(c, k, v)
        self.cfg = c
        self._h = ctypes.c_void_p()
        self._lib = lib
        check(lib.sert_create(ctypes.byref(c), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.sert_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # tensors
    def tensor_size(self, which):
        return int(self._lib.sert_tensor_size(self._h, which))

    def set_tensor(self, which, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        check(self._lib.sert_set_tensor(self._h, which, a.ctypes.data, a.size))

    def get_tensor(self, which, shape=None, out=None):
        n = self.tensor_size(which)
        if out is None:
            out = np.empty(n, dtype=np.float32)
        assert out.dtype == np.float32 and out.size == n and out.flags['C_CONTIGUOUS']
        check(self._lib.sert_get_tensor(self._h, which, out.ctypes.data, n))
        return out.reshape(shape) if shape is not None else out

    def set_step(self, t):
        check(self._lib.sert_set_step(self._h, int(t)))

    def get_step(self):
        return int(self._lib.sert_get_step(self._h))

    def set_eval_draws(self, n):
        check(self._lib.sert_set_eval_draws(self._h, int(n)))

    def get_eval_draws(self):
        return int(self._lib.sert_get_eval_draws(self._h))

    def negatives_of_step(self, position, evaluation=False):
        """(B, z) int64: the negatives the device sampler draws at training position `position` (= get_step() before the
        step) or for the `position`-th evaluated batch."""
        out = np.empty((self.cfg.batch_size, self.cfg.num_negatives), dtype=np.int64)
        check(self._lib.sert_negatives_of_step(self._h, int(position), int(bool(evaluation)), out.ctypes.data))
        return out

    # data
    def upload_dataset(self, split, x, y_int=None, csr=None, w=None):
        x = np.ascontiguousarray(x)
        n = x.shape[0]
        keep = [x]
        y_addr = ip = ix = da = None
        if y_int is not None:
            y_int = np.ascontiguousarray(y_int, dtype=np.int32)
            keep.append(y_int)
            y_addr = y_int.ctypes.data
        if csr is not None:
            indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
            indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
            data = np.ascontiguousarray(csr.data, dtype=np.float32)
            keep += [indptr, indices, data]
            ip, ix, da = indptr.ctypes.data, indices.ctypes.data, data.ctypes.data
        w_addr = None
        if w is not None:
            w = np.ascontiguousarray(w, dtype=np.float32)
            keep.append(w)
            w_addr = w.ctypes.data
        check(self._lib.sert_upload_dataset(self._h, split, x.ctypes.data if n else None,
                                            y_addr, ip, ix, da, w_addr, n))
        del keep

    # hot path
    def train_batch(self, batch_index, negatives=None):
        loss = ctypes.c_float()
        if negatives is not None:
            negatives = np.ascontiguousarray(negatives, dtype=np.int64)
        check(self._lib.sert_train_batch(self._h, int(batch_index), _addr(negatives),
                                         ctypes.byref(loss)))
        return np.float32(loss.value)

    def hint_next_batch(self, next_batch_index):
        """The batch that will be trained after the next train_batch call (or None):
        its parameter-only forward part is enqueued before that call waits for its loss."""
        check(self._lib.sert_hint_next_batch(
            self._h, -1 if next_batch_index is None else int(next_batch_index)))

    def train_batches(self, batch_indices):
        idx = np.ascontiguousarray(batch_indices, dtype=np.int64)
        out = np.empty(idx.size, dtype=np.float32)
        check(self._lib.sert_train_batches(self._h, idx.ctypes.data, idx.size, out.ctypes.data))
        return out

    def eval_batch(self, split, batch_index, negatives=None):
        loss = ctypes.c_float()
        if negatives is not None:
            negatives = np.ascontiguousarray(negatives, dtype=np.int64)
        check(self._lib.sert_eval_batch(self._h, split, int(batch_index), _addr(negatives),
                                        ctypes.byref(loss)))
        return np.float32(loss.value)

    def eval_batches(self, split, batch_indices):
        """Unweighted batch-mean losses of several batches, one host synchronisation."""
        idx = np.ascontiguousarray(batch_indices, dtype=np.int64)
        out = np.empty(idx.shape[0], dtype=np.float32)
        if idx.shape[0]:
            check(self._lib.sert_eval_batches(self._h, split, idx.ctypes.data, idx.shape[0], out.ctypes.data))
        return out

    def predict_project(self, avg):
        avg = np.ascontiguousarray(avg, dtype=np.float32)
        q = avg.shape[0]
        out = np.empty((q, self.cfg.entity_dim), dtype=np.float32)
        check(self._lib.sert_predict_project(self._h, avg.ctypes.data, q, out.ctypes.data))
        return out

    def predict_tokens(self, ids):
        ids = np.ascontiguousarray(ids)
        assert ids.dtype.itemsize == self.cfg.id_bytes
        rows = ids.shape[0]
        out = np.empty((rows, self.cfg.window_size, self.cfg.num_entities), dtype=np.float32)
        check(self._lib.sert_predict_tokens(self._h, ids.ctypes.data, rows, out.ctypes.data))
        return out

    # data parallel
    def comm_init(self, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        check(self._lib.sert_comm_init(self._h, unique_id, rank, world))

    def comm_init_host(self, rank, world, alltoall):
        """Host-mediated exchange (verification transport, sert_comm_init_host):
        ``alltoall(send, send_offsets, send_counts, recv, recv_offsets, recv_counts)`` moves float32
        segments between the ranks (numpy views of the engine's pinned buffers; offsets and counts are
        int64 arrays of length world, in floats) -- see sert_amd.distributed.host_alltoall."""
        def trampoline(_user, send, soff, scnt, recv, roff, rcnt):
            try:
                so = np.ctypeslib.as_array(soff, shape=(world,))
                sc = np.ctypeslib.as_array(scnt, shape=(world,))
                ro = np.ctypeslib.as_array(roff, shape=(world,))
                rc = np.ctypeslib.as_array(rcnt, shape=(world,))
                ns = int((so + sc).max()) if world else 0
                nr = int((ro + rc).max()) if world else 0
                sbuf = np.ctypeslib.as_array(send, shape=(max(ns, 1),))
                rbuf = np.ctypeslib.as_array(recv, shape=(max(nr, 1),))
                alltoall(sbuf, so, sc, rbuf, ro, rc)
                return 0
            except Exception:  # noqa: BLE001 - reported through the C error path
                import traceback
                traceback.print_exc()
                return 1
        self._host_alltoall = ALLTOALL_FN(trampoline)   # keep the thunk alive
        check(self._lib.sert_comm_init_host(self._h, rank, world, self._host_alltoall, None))

    def comm_stats(self):
        """sert_comm_stats as a dict (world 1 / no communicator: exchange 'none')."""
        v = (ctypes.c_double * 8)()
        check(self._lib.sert_comm_stats(self._h, v, 8))
        return {'world': int(v[0]), 'exchange': {0: 'none', 1: 'zero1', 2: 'rows'}[int(v[1])],
                'bytes_per_step': float(v[2]), 'zero1_bytes_per_step': float(v[3]),
                'transport': {0: 'none', 1: 'rccl', 2: 'host'}[int(v[4])], 'steps': int(v[5]),
                'rows_fetched_per_batch': float(v[6]), 'rows_served_per_batch': float(v[7])}

    # diagnostics
    def synchronize(self):
        check(self._lib.sert_synchronize(self._h))

    def timing_enable(self, on=True):
        """False / 0: off; True / 1: every kernel group alone on one queue; 2: in the step (see sert_hip.h)."""
        check(self._lib.sert_timing_enable(self._h, int(on)))

    def timing_launches(self):
        """In-step mode: timed launches per training step and group."""
        n = self._lib.sert_timing_count(self._h)
        return {self._lib.sert_timing_name(self._h, i).decode():
                self._lib.sert_timing_launches(self._h, i) for i in range(n)}

    def timing_reset(self):
        check(self._lib.sert_timing_reset(self._h))

    def poison_scratch(self):
        """sert_debug_poison_scratch (test hook): NaNs into the gradient scratch, wrong run bounds behind it."""
        self._lib.sert_debug_poison_scratch.argtypes = [ctypes.c_void_p]
        check(self._lib.sert_debug_poison_scratch(self._h))

    def update_counts(self):
        """sert_debug_update_counts (test hook): launches of the word-table update by kernel form since creation."""
        v = (ctypes.c_int64 * 10)()
        self._lib.sert_debug_update_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
        check(self._lib.sert_debug_update_counts(self._h, v, 10))
        names = ('dense', 'lazy', 'skip_32_1', 'skip_64_1', 'skip_32_3', 'skip_64_2', 'skip_64_3', 'skip_64_4', 'skip_full', 'skip_sparse')
        return dict(zip(names, (int(x) for x in v)))

    def timings(self):
        n = self._lib.sert_timing_count(self._h)
        return {self._lib.sert_timing_name(self._h, i).decode():
                self._lib.sert_timing_avg_us(self._h, i) for i in range(n)}


def comm_unique_id():
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    check(load().sert_comm_unique_id(buf))
    return buf.raw


def score_topk(entities, projections, k, device=0):
    """Batched VectorSpaceCallback scoring (bin/query.py:239-370)."""
    e = np.ascontiguousarray(entities, dtype=np.float32)
    p = np.ascontiguousarray(projections, dtype=np.float32)
    assert e.ndim == 2 and p.ndim == 2 and e.shape[1] == p.shape[1]
    q = p.shape[0]
    idx = np.empty((q, k), dtype=np.int32)
    val = np.empty((q, k), dtype=np.float32)
    check(load().sert_score_topk(device, e.ctypes.data, e.shape[0], e.shape[1], p.ctypes.data, q,
                                 k, idx.ctypes.data, val.ctypes.data))
    return idx, val


class PinnedBuffer(object):
    """Page-locked host memory (sert_host_alloc) viewed as a numpy array; grows on demand."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        self._ptr = ctypes.c_void_p()
        self._bytes = 0
        self._lib = load()

    def view(self, shape):
        if os.environ.get('SERT_NO_PINNED'):     # cross-check knob: ordinary pageable arrays
            return np.empty(shape, dtype=self.dtype)
        need = int(np.prod(shape)) * self.dtype.itemsize
        if need > self._bytes:
            self.free()
            cap = max(need, 1 << 16)
            check(self._lib.sert_host_alloc(ctypes.byref(self._ptr), cap))
            self._bytes = cap
        raw = (ctypes.c_char * max(need, 1)).from_address(self._ptr.value)
        return np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def holds(self, array):
        """Is `array` a view of this buffer (so that no staging copy is needed)?"""
        if not self._ptr.value or not isinstance(array, np.ndarray) or not array.flags['C_CONTIGUOUS']:
            return False
        addr = array.ctypes.data
        return self._ptr.value <= addr and addr + array.nbytes <= self._ptr.value + self._bytes

    def free(self):
        if self._ptr.value:
            self._lib.sert_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()
            self._bytes = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scorer(object):
    """Persistent device copy of the (L2-normalised) entity table + top-k scoring
    (sert_scorer_* in include/sert_hip.h).

    Host arrays cross the boundary through page-locked buffers the scorer owns: fill
    ``query_buffer(Q)`` in place (or pass any array: it is copied there) and read the result
    views, which stay valid until the next call on this scorer."""

    #: upper bound on Q_chunk * V_e floats materialised on the host when no device top-k applies
    MAX_HOST_SCORES = 1 << 26

    def __init__(self, entities, device=0):
        e = np.ascontiguousarray(entities, dtype=np.float32)
        assert e.ndim == 2
        self.num_entities, self.dim = e.shape
        self._lib = load()
        self._h = ctypes.c_void_p()
        check(self._lib.sert_scorer_create(device, e.ctypes.data, e.shape[0], e.shape[1],
                                           ctypes.byref(self._h)))
        self._pin_q, self._pin_idx, self._pin_val = PinnedBuffer(np.float32), PinnedBuffer(np.int32), PinnedBuffer(np.float32)

    def query_buffer(self, num_queries):
        """A page-locked (Q, d) float32 array to build the query block in (zero-copy upload)."""
        return self._pin_q.view((num_queries, self.dim))

    def _stage(self, projections):
        p = np.asarray(projections, dtype=np.float32)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        assert p.shape[1] == self.dim
        if self._pin_q.holds(p):
            return p
        staged = self._pin_q.view(p.shape)
        np.copyto(staged, p)
        return staged

    def topk(self, projections, k):
        p = self._stage(projections)
        q = p.shape[0]
        idx = self._pin_idx.view((q, k))
        val = self._pin_val.view((q, k))
        check(self._lib.sert_scorer_topk(self._h, p.ctypes.data, q, k, idx.ctypes.data,
                                         val.ctypes.data))
        return idx, val

    def scores(self, projections):
        p = np.ascontiguousarray(projections, dtype=np.float32)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        assert p.shape[1] == self.dim
        out = np.empty((p.shape[0], self.num_entities), dtype=np.float32)
        check(self._lib.sert_scorer_scores(self._h, p.ctypes.data, p.shape[0], out.ctypes.data))
        return out

    def rank(self, projections, k=None):
        """(idx, score) per query, best first; k=None ranks every entity."""
        if k is not None and k <= min(self.num_entities, 1024):
            return self.topk(projections, k)
        # no device top-k for this k: full score rows, a bounded number of queries at a time
        # (the reference scores one query at a time, query.py:304-318; 10k queries x 100k
        # entities in one block would be 4 GB of host memory)
        p = np.asarray(projections, dtype=np.float32)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        keep = self.num_entities if k is None else min(k, self.num_entities)
        step = max(1, self.MAX_HOST_SCORES // self.num_entities)
        idx = np.empty((p.shape[0], keep), dtype=np.int32)
        val = np.empty((p.shape[0], keep), dtype=np.float32)
        for lo in range(0, p.shape[0], step):
            sc = self.scores(p[lo:lo + step])
            order = np.argsort(-sc, axis=1, kind='stable')[:, :keep]   # ties: lowest index first
            idx[lo:lo + step] = order
            val[lo:lo + step] = np.take_along_axis(sc, order, axis=1)
        return idx, val

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.sert_scorer_destroy(self._h)
            self._h = None
        for b in (getattr(self, '_pin_q', None), getattr(self, '_pin_idx', None), getattr(self, '_pin_val', None)):
            if b is not None:
                b.free()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bench_gemm(M, N, K, ta=0, tb=0, epi=0, splits=1, iters=20, device=0):
    """Average launch time (us) of the fp32 MFMA GEMM on random device operands."""
    us = ctypes.c_double()
    check(load().sert_bench_gemm(device, ta, tb, epi, M, N, K, splits, iters, ctypes.byref(us)))
    return us.value


MEMBENCH_COPY, MEMBENCH_READ, MEMBENCH_GATHER, MEMBENCH_OPTIMIZER = 0, 1, 2, 3
SEPARATE_ALLOCATIONS = ctypes.c_size_t(-1).value


def debug_word_index_sum(ids, vocab, src, batch=0, row_groups=1, dense_heavy=False, divisor=1.0, sort_level0=False):
    """Host only: the word-table gradient of one batch through the inverted index as the segmented-sum kernels walk it
    (sert_debug_word_index_sum).  ids (num_batches, B, n) unsigned; src (B, d) float32.  sort_level0: level 0 sorted by item
    length with every item's first row number in its descriptor (what the vectorspace models upload).  Returns (grad (vocab, d), stats)."""
    lib = load()
    ids = np.ascontiguousarray(ids)
    assert ids.ndim == 3 and ids.dtype in (np.uint8, np.uint16, np.uint32), (ids.shape, ids.dtype)
    src = np.ascontiguousarray(src, dtype=np.float32)
    nb, B, n = ids.shape
    assert src.shape[0] == B
    d = src.shape[1]
    out = np.empty((vocab, d), dtype=np.float32)
    stats = np.zeros(8, dtype=np.int64)
    lib.sert_debug_word_index_sum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_void_p, ctypes.c_void_p]
    check(lib.sert_debug_word_index_sum(_addr(ids), ids.dtype.itemsize, nb, B, n, int(vocab), int(row_groups), int(bool(dense_heavy)) | (2 if sort_level0 else 0),
                                        int(batch), _addr(src), d, float(divisor), _addr(out), _addr(stats)))
    keys = ('levels', 'items', 'partial_rows', 'final_items', 'dense_words', 'row_groups', 'level0_items', 'distinct_words')
    return out, dict(zip(keys, [int(x) for x in stats]))


def debug_gemm(A, B, ta=0, tb=0, epi=0, bias=None, device=0):
    """C = epi(op(A).op(B)) through the library's GEMM dispatch (sert_debug_gemm); shapes from the arrays."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    K, M = A.shape if ta else A.shape[::-1]
    N = B.shape[0] if tb else B.shape[1]
    assert (B.shape[1] if tb else B.shape[0]) == K, (A.shape, B.shape)
    C = np.empty((M, N), dtype=np.float32)
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=np.float32)
    lib.sert_debug_gemm.argtypes = [ctypes.c_int] * 7 + [ctypes.c_void_p] * 4
    check(lib.sert_debug_gemm(device, int(ta), int(tb), int(epi), M, N, K, _addr(A), _addr(B),
                              _addr(bias) if bias is not None else None, _addr(C)))
    return C


def debug_gemm_splitk(A, B, splits, device=0):
    """(A^T.B, column sums of B) through the split-K launch + combine of a training step (sert_debug_gemm_splitk);
    A (K, M), B (K, N)."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    K, M = A.shape
    N = B.shape[1]
    assert B.shape[0] == K
    out = np.empty(M * N + N, dtype=np.float32)
    lib.sert_debug_gemm_splitk.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    check(lib.sert_debug_gemm_splitk(device, M, N, K, int(splits), _addr(A), _addr(B), _addr(out)))
    return out[:M * N].reshape(M, N), out[M * N:]


def debug_gemm_longk(A, B, splits, tb=0, device=0):
    """A.op(B) over a long K cut into `splits` k ranges + combine (sert_debug_gemm_longk); A (M, K), B (K, N) or (N, K) if tb."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    M, K = A.shape
    N = B.shape[0] if tb else B.shape[1]
    assert (B.shape[1] if tb else B.shape[0]) == K
    out = np.empty((M, N), dtype=np.float32)
    lib.sert_debug_gemm_longk.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
    check(lib.sert_debug_gemm_longk(device, int(tb), M, N, K, int(splits), _addr(A), _addr(B), _addr(out)))
    return out


def bench_memory(kind, nbytes, table_bytes=0, row_bytes=512, window=10, gap_bytes=0, blocks=0, iters=20, device=0):
    """Average launch time (us) of a memory-system micro-benchmark (include/sert_hip.h:
    sert_bench_memory): stream copy / read, the window gather over random rows, the dense
    optimiser's seven streams."""
    us = ctypes.c_double()
    check(load().sert_bench_memory(device, kind, nbytes, table_bytes, row_bytes, window, gap_bytes, blocks, iters,
                                   ctypes.byref(us)))
    return us.value


def debug_row_lists(allbits, rank, rows_per_rank, vocab, batch):
    """sert_debug_row_lists (host only): the exchange lists of one rank and batch as numpy arrays."""
    allbits = np.ascontiguousarray(allbits, dtype=np.uint32)
    world, nb, bw = allbits.shape
    cap = int(vocab) + 2
    i32 = lambda n: np.zeros(n, dtype=np.int32)
    scnt, fcnt = i32(world), i32(world)
    srows, frows, urows, ptr, ent = (i32(cap * world) for _ in range(5))
    sizes = np.zeros(5, dtype=np.int64)
    check(load().sert_debug_row_lists(allbits.ctypes.data, world, rank, nb, bw, rows_per_rank, vocab, batch,
                                      scnt.ctypes.data, fcnt.ctypes.data, srows.ctypes.data, frows.ctypes.data,
                                      urows.ctypes.data, ptr.ctypes.data, ent.ctypes.data, cap * world, sizes.ctypes.data))
    ns, nf, nu, ne = (int(x) for x in sizes[:4])
    return dict(serve_cnt=scnt, fetch_cnt=fcnt, serve_rows=srows[:ns], fetch_rows=frows[:nf], union_rows=urows[:nu],
                ptr=ptr[:nu + 1], ent=ent[:ne], max_xfer_rows=int(sizes[4]))


class profile_range(object):
    """with profile_range('epoch 3'): ...   -- a roctx range (sert_profile_range_push / pop)."""

    def __init__(self, name):
        self.name = name.encode() if isinstance(name, str) else name

    def __enter__(self):
        check(load().sert_profile_range_push(self.name))
        return self

    def __exit__(self, *exc):
        check(load().sert_profile_range_pop())
