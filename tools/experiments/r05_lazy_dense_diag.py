"""Where do the lazy and the dense run of tests/test_gpu_lazy_dense.py part?  Per step: rows of R_w / state that differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sert_amd import _capi as C
from tests import util as U
from tests import test_gpu_lazy_dense as T

kind = sys.argv[1] if len(sys.argv) > 1 else 'loglinear'
B, n, Vw, d = T.B, T.n, T.Vw, T.d
rng = np.random.RandomState(97)
nb = len(T.KINDS)
if kind == 'vectorspace':
    p = U.make_vs_problem(97, nb * B, n, 4, Vw, 12, d, d)
    mk = lambda keep: U.vs_engine(p, B, n, 4, 0.05, keep_grads=keep)
else:
    p = U.make_ll_problem(97, nb * B, n, Vw, 24, d, 'int')
    mk = lambda keep: U.ll_engine(p, B, n, 0.05, keep_grads=keep)
p['X'] = T._tokens(rng, T.KINDS)
order = list(range(nb)) + [1, 2, 0, 4, 3]
negs = [rng.randint(0, 12, (B, 4)).astype(np.int64) for _ in order] if kind == 'vectorspace' else None
engs = [mk(1), mk(0)]
for e in engs:
    e.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
for s, b in enumerate(order):
    ls = []
    for e in engs:
        e.hint_next_batch(order[s + 1] if s + 1 < len(order) else None)
        ls.append(e.train_batch(b, negs[s]) if negs else e.train_batch(b))
    line = 'step %2d batch %d (%s) next %s loss_equal %s' % (s + 1, b, T.KINDS[b], T.KINDS[order[s + 1]] if s + 1 < len(order) else '-', ls[0] == ls[1])
    for name, t in (('RW', C.T_RW), ('S0', C.T_STATE0_RW), ('S1', C.T_STATE1_RW), ('W', C.T_W), ('B', C.T_B)):
        a, c = engs[0].get_tensor(t), engs[1].get_tensor(t)
        if name in ('RW', 'S0', 'S1'):
            a, c = a.reshape(Vw, d), c.reshape(Vw, d)
            bad = np.where((a != c).any(axis=1))[0]
            touched = np.unique(p['X'][b * B:(b + 1) * B])
            line += ' | %s rows differ %d (touched among them %d)' % (name, len(bad), len(np.intersect1d(bad, touched)))
            if len(bad) and name == 'RW':
                r = bad[0]
                line += ' e.g. row %d max|diff| %.3g' % (r, np.abs(a[r] - c[r]).max())
        else:
            line += ' | %s differ %d' % (name, int((a != c).sum()))
    print(line, flush=True)
