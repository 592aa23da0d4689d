"""Which of the role-specialised GEMMs is off, and for which shapes?  (SERT_STRIP_GEMM=2)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util as U
from oracle import sert_oracle as O
from sert_amd import _capi as C
for (B, dw, de) in ((1030, 96, 64), (1024, 96, 64), (1030, 128, 64), (1030, 64, 128), (1030, 96, 96), (1030, 128, 128), (2048, 32, 32)):
    n, z, Vw, Ve = 2, 3, 500, 40
    p = U.make_vs_problem(0, B, n, z, Vw, Ve, dw, de, zipf=True)
    outs = []
    for rep in range(3):
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        neg = np.random.RandomState(5).randint(0, Ve, size=(B, z)).astype(np.int64)
        loss = eng.train_batch(0, neg)
        T = eng.get_tensor(C.T_ACT_T, (B, de)).copy()
        DH = eng.get_tensor(C.T_ACT_DH, (B, dw)).copy()
        outs.append((loss, T, DH))
        eng.close()
    ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
    loss_ref, grads_ref, f = ora.loss_and_grads(p['X'], p['y'], p['w'], neg)
    bad_T = np.argwhere(np.abs(outs[0][1] - f['t']) > 1e-4)
    bad_D = np.argwhere(np.abs(outs[0][2] - f['dh']) > 1e-4 * np.abs(f['dh']).max())
    print('B=%d dw=%d de=%d  loss %.6f ref %.6f | T err %.2e (%d bad, rows %s cols %s) | DH err %.2e (%d bad, rows %s cols %s) | repeatable T %s DH %s' % (
        B, dw, de, outs[0][0], loss_ref, U.rel_err(outs[0][1], f['t']), len(bad_T),
        sorted(set(bad_T[:, 0]))[:6], sorted(set(bad_T[:, 1]))[:6],
        U.rel_err(outs[0][2], f['dh']), len(bad_D), sorted(set(bad_D[:, 0]))[:6], sorted(set(bad_D[:, 1]))[:6],
        all(np.array_equal(outs[0][1], o[1]) for o in outs), all(np.array_equal(outs[0][2], o[2]) for o in outs)))
