"""Build + ctypes binding of oracle/sert_cpu.c, the multithreaded CPU restatement of the LSE
training step and of batched cosine scoring.

TEST / MEASUREMENT INFRASTRUCTURE ONLY: used by ``bench.py``'s ``cpu_baseline`` leg, by
``tests/test_cpu_baseline.py`` (which pins it to the numpy oracle) and built by
``__graft_entry__.build()``.  Nothing under ``sert_amd/`` may import it.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'sert_cpu.c')
OUT_DIR = os.path.join(HERE, '_build')


def build(native=False, out_dir=None, force=False):
    """gcc -O3 -fopenmp -> libsert_cpu[_native].so.  ``native``: -march=native (the bench builds
    that on the node it runs on); otherwise x86-64-v3 (AVX2 + FMA), which runs on any recent
    host, so that the prebuilt file can travel."""
    out_dir = out_dir or OUT_DIR
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, 'libsert_cpu%s.so' % ('_native' if native else ''))
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(SRC):
        return lib
    cmd = ['gcc', '-O3', '-march=native' if native else '-march=x86-64-v3', '-fopenmp', '-fno-math-errno',
           '-shared', '-fPIC', SRC, '-o', lib, '-lm']
    subprocess.check_call(cmd)
    return lib


def load(native=False, out_dir=None):
    lib = ctypes.CDLL(build(native=native, out_dir=out_dir))
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.sert_cpu_create.restype = vp
    lib.sert_cpu_create.argtypes = [ci] * 7 + [cf] + [vp] * 4
    lib.sert_cpu_destroy.argtypes = [vp]
    lib.sert_cpu_threads.argtypes = [vp]
    lib.sert_cpu_phases.argtypes = [vp, vp]
    lib.sert_cpu_get.argtypes = [vp] * 5
    lib.sert_cpu_train_step.restype = cf
    lib.sert_cpu_train_step.argtypes = [vp, ci] + [vp] * 4
    lib.sert_cpu_index_batch.argtypes = [vp, ci, vp]
    lib.sert_cpu_score_topk.argtypes = [vp, ci, ci, vp, ci, ci, vp]
    return lib


class VectorSpaceCPU(object):
    """Same constructor arguments and train_step signature as oracle.sert_oracle.VectorSpaceOracle."""

    def __init__(self, batch_size, window_size, num_negative_samples, R_w, R_e, W, b,
                 regularization_lambda, native=False, out_dir=None):
        self.lib = load(native, out_dir)
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        R_w, R_e, W, b = f(R_w), f(R_e), f(W), f(b)
        self.shapes = (R_w.shape, R_e.shape, W.shape, b.shape)
        assert max(R_w.shape[1], R_e.shape[1]) <= 512
        self.B, self.n, self.z = batch_size, window_size, num_negative_samples
        self.h = self.lib.sert_cpu_create(batch_size, window_size, num_negative_samples, R_w.shape[0],
                                          R_e.shape[0], R_w.shape[1], R_e.shape[1], regularization_lambda,
                                          R_w.ctypes.data, R_e.ctypes.data, W.ctypes.data, b.ctypes.data)
        self.threads = self.lib.sert_cpu_threads(self.h)

    def index_batch(self, slot, X):
        """Word -> positions index of a batch (a static slice of the data set), kept in `slot`:
        built once, outside any timed region, as the GPU engine does at upload."""
        X = np.ascontiguousarray(X, dtype=np.int32)
        assert X.shape == (self.B, self.n)
        assert self.lib.sert_cpu_index_batch(self.h, slot, X.ctypes.data) == 0

    def train_step(self, X, y, w, neg, slot=None):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        X, y, neg = i32(X), i32(y), i32(neg)
        w = np.ascontiguousarray(w, dtype=np.float32)
        assert X.shape == (self.B, self.n) and neg.shape == (self.B, self.z)
        if slot is None:
            slot = 0
            self.index_batch(0, X)
        return float(self.lib.sert_cpu_train_step(self.h, slot, X.ctypes.data, y.ctypes.data, w.ctypes.data,
                                                  neg.ctypes.data))

    def phases_ms(self):
        """Wall time of the last step by phase: forward + NCE + dh, dW, entity grouping, segmented
        sums, L2 + Adam, total."""
        out = np.zeros(6, dtype=np.float64)
        self.lib.sert_cpu_phases(self.h, out.ctypes.data)
        return dict(zip(('forward_nce_dh', 'dW', 'entity_grouping', 'segmented_sums', 'l2_adam', 'total'), out.tolist()))

    def params(self):
        out = [np.empty(s, dtype=np.float32) for s in self.shapes]
        self.lib.sert_cpu_get(self.h, *[a.ctypes.data for a in out])
        return dict(R_w=out[0], R_e=out[1], W=out[2], b=out[3])

    def close(self):
        if self.h:
            self.lib.sert_cpu_destroy(self.h)
            self.h = None

    __del__ = close


def score_topk(entities, projections, k, native=False, out_dir=None):
    """Indices of the k highest cosines per query (ties: lowest entity index)."""
    lib = load(native, out_dir)
    E = np.ascontiguousarray(entities, dtype=np.float32)
    P = np.ascontiguousarray(projections, dtype=np.float32)
    E = E / np.linalg.norm(E, axis=1, keepdims=True)
    P = P / np.linalg.norm(P, axis=1, keepdims=True)
    idx = np.empty((P.shape[0], k), dtype=np.int32)
    lib.sert_cpu_score_topk(E.ctypes.data, E.shape[0], E.shape[1], P.ctypes.data, P.shape[0], k, idx.ctypes.data)
    return idx


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when
    unlimited: more busy threads than that get throttled in 100 ms periods."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            return max(1, int(int(quota) / float(period)))
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            quota = int(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            period = int(f.read())
        if quota > 0:
            return max(1, int(quota / float(period)))
    except (OSError, ValueError):
        pass
    return None


def one_socket_cores():
    """Logical CPUs of socket 0 this process may use, one per physical core, capped by the
    container's CPU quota."""
    allowed = sorted(os.sched_getaffinity(0))
    picked, seen = [], set()
    for cpu in allowed:
        base = '/sys/devices/system/cpu/cpu%d/topology/' % cpu
        try:
            with open(base + 'physical_package_id') as f:
                pkg = int(f.read())
            with open(base + 'core_id') as f:
                core = int(f.read())
        except (OSError, ValueError):
            pkg, core = 0, cpu
        if pkg != 0 or core in seen:
            continue
        seen.add(core)
        picked.append(cpu)
    picked = picked or allowed
    quota = cpu_quota()
    if quota is not None:
        picked = picked[:max(1, quota)]
    return picked
