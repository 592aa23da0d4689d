import sys; sys.path.insert(0,'.')
import numpy as np
from tests import util as U
from tests.util import C
B, n, z, Vw, Ve, de, dw = 32, 3, 4, 5000, 12, 16, 16
p = U.make_vs_problem(41, B * 4, n, z, Vw, Ve, dw, de)
neg = p['rng'].randint(0, Ve, (B, z)).astype(np.int64)
outs = []
for keep in (1, 0):
    eng = U.vs_engine(p, B, n, z, 0.05, keep_grads=keep)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    res=[]
    for s in range(3):
        l = eng.train_batch(s % 4, neg)
        res.append((l, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_STATE0_RW).copy(), eng.get_tensor(C.T_STATE1_RW).copy()))
    outs.append(res)
    eng.close()
for s in range(3):
    a,b = outs[0][s], outs[1][s]
    for name,i in (('p',1),('m',2),('v',3)):
        d = np.abs(a[i]-b[i]); nz = np.count_nonzero(d)
        print('step',s,name,'max abs diff %.3e'%d.max(), 'n diff', nz, 'rel %.2e' % (d.max()/max(1e-30,np.abs(a[i]).max())))
