"""Per-wave timeline of workgroup 0 of the LAST gemm_roles_nn launch (variant build -DSR_TIMELINE,
SERT_LIB=.../libsert_TL.so SERT_STRIP_GEMM=2): shader-clock stamps at the start of every iteration
(after the barrier) and at the end of the wave's work (before the next barrier)."""
import ctypes
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util as U
from sert_amd import _capi as C

B, n, z, Vw, Ve, d = 65536, 10, 10, 100000, 1000, 128
p = U.make_vs_problem(3, B, n, z, Vw, Ve, d, d, zipf=True)
eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0, seed=11)
eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
for s in range(3):
    eng.train_batch(0)
lib = ctypes.CDLL(os.environ['SERT_LIB'])
buf = np.zeros(8 * 16 * 2, dtype=np.uint64)
rc = lib.sert_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
t = buf.reshape(8, 16, 2).astype(np.int64)
t0 = t[:, 0, 0].min()
names = ['compute0', 'compute1', 'compute2', 'compute3', 'loader4', 'loader5', 'epilogue6', 'epilogue7']
print('cycles relative to the first stamp; the last launch with this kernel is the dh GEMM (no tanh)')
for w in (0, 4, 6):
    print(names[w])
    for it in range(10):
        print('   it %2d  start %7d  work %6d  (to next start %6d)' % (
            it, t[w, it, 0] - t0, t[w, it, 1] - t[w, it, 0], (t[w, it + 1, 0] - t[w, it, 0]) if it < 9 else -1))
