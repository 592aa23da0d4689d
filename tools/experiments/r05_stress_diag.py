"""Where the word table of the failing stress shape differs from the oracle (argv: KEY=VALUE environment settings)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for a in sys.argv[1:]:
    k, v = a.split('=', 1); os.environ[k] = v
sys.path.insert(0, ROOT)
import numpy as np
from tests import util as U
from sert_amd import _capi as C
from oracle import sert_oracle as O
dims = {'B': 129, 'n': 2, 'z': 17, 'Vw': 70000, 'Ve': 2, 'dw': 128, 'de': 128}
B, n, z = dims['B'], dims['n'], dims['z']
steps = int(os.environ.get('STEPS', 3))
p = U.make_vs_problem(0, B * steps, n, z, dims['Vw'], dims['Ve'], dims['dw'], dims['de'], zipf=True)
eng = U.vs_engine(p, B, n, z, 0.01)
eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
for s in range(steps):
    sl = slice(s * B, (s + 1) * B)
    neg = p['rng'].randint(0, dims['Ve'], size=(B, z)).astype(np.int64)
    loss_ref, grads_ref, f = ora.loss_and_grads(p['X'][sl], p['y'][sl], p['w'][sl], neg)
    ora.opt.update(ora.params(), grads_ref)
    loss = eng.train_batch(s, neg)
    got = eng.get_tensor(C.T_RW, (dims['Vw'], dims['dw']))
    ref = ora.R_w
    d = np.abs(got - ref)
    rows = np.where(d.max(axis=1) > 1e-6)[0]
    touched_now = np.unique(p['X'][sl])
    touched_all = np.unique(p['X'][:(s + 1) * B])
    print('step', s, 'loss', loss, loss_ref, 'rel_err Rw', U.rel_err(got.ravel(), ref.ravel()), 'max abs', d.max(), 'rows off', len(rows),
          'of which touched this step', len(np.intersect1d(rows, touched_now)), 'touched so far', len(np.intersect1d(rows, touched_all)),
          'first', rows[:8], 'max|ref|', np.abs(ref).max())
    for t, name in ((C.T_STATE0_RW, 'm'), (C.T_STATE1_RW, 'v')):
        try:
            st = eng.get_tensor(t, (dims['Vw'], dims['dw']))
            o = ora.opt.m[0] if name == 'm' else ora.opt.v[0]
        except Exception as e:
            continue
eng.close()
