"""Loglinear at C4's table sizes (V_w 500k, V_e 100k, d 300), batch 1024, Zipf tokens as in bench.py: two steps with every
GEMM on the bf16 pipe (gemm_x3.h: logits, dW and the long-K dG in k ranges) against the same steps with SERT_GEMM_FP32=1 --
losses, gradients of step 1 (keep_grads) and parameters after step 2."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sert_amd import _capi as C  # noqa: E402
from tests import util as U  # noqa: E402
from oracle import sert_oracle as O  # noqa: E402

B, n, Vw, Ve, d = 1024, 10, 500000, 100000, 300
rng = np.random.RandomState(0)
X, y, w = bench.synth_data(rng, 2 * B, n, Vw, Ve)
p = dict(Rw=O.glorot_uniform(rng, (Vw, d)), W=O.glorot_uniform(rng, (d, Ve)), b=(0.1 * rng.randn(Ve)).astype(np.float32), X=X)


def run(fp32):
    os.environ['SERT_GEMM_FP32'] = '1' if fp32 else '0'
    eng = U.ll_engine(p, B, n, 0.01, keep_grads=1)
    eng.upload_dataset(C.SPLIT_TRAIN, X, y_int=y.astype(np.int32), w=w)
    out = {'loss': []}
    out['loss'].append(eng.train_batch(0))
    for name, t, shape in (('gRw', C.T_GRAD_RW, (Vw, d)), ('gW', C.T_GRAD_W, (d, Ve)), ('gb', C.T_GRAD_B, (1, Ve))):
        out[name] = eng.get_tensor(t, shape).copy()
    out['loss'].append(eng.train_batch(1))
    for name, t, shape in (('Rw', C.T_RW, (Vw, d)), ('W', C.T_W, (d, Ve)), ('b', C.T_B, (1, Ve))):
        out[name] = eng.get_tensor(t, shape).copy()
    eng.close()
    return out


a, b = run(False), run(True)
print('losses bf16 pipe', a['loss'], ' fp32 MFMA', b['loss'])
for k in ('gRw', 'gW', 'gb', 'Rw', 'W', 'b'):
    den = np.abs(b[k]).max()
    print('%-4s max |diff| / max |ref| = %.3e   rows touched %d' % (k, np.abs(a[k] - b[k]).max() / den, int((np.abs(b[k]).sum(axis=1) > 0).sum())))
dWa, dWb = a['gW'], b['gW']
err = np.abs(dWa - dWb).max(axis=0)
bad = np.nonzero(err > 1e-4 * np.abs(dWb).max())[0]
print('bad columns:', len(bad), 'first', bad[:10], 'last', bad[-10:] if len(bad) else None)
if len(bad):
    print('bad mod 160 histogram:', np.bincount(bad % 160, minlength=160).nonzero()[0][:40])
    errr = np.abs(dWa - dWb).max(axis=1)
    print('bad rows:', np.nonzero(errr > 1e-4 * np.abs(dWb).max())[0][:20])
print('--- repeatability: the same path twice')
for fp32 in (False, True):
    r1, r2 = run(fp32), run(fp32)
    print('fp32 MFMA' if fp32 else 'bf16 pipe', {k: float(np.abs(r1[k] - r2[k]).max() / np.abs(r2[k]).max()) for k in ('gRw', 'gW', 'gb', 'Rw', 'W', 'b')})
