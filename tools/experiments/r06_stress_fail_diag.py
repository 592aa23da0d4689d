"""The two shapes the round-6 stress runs (seeds 31, 32) flagged: the HIP engine and the float32 oracle against the SAME oracle in float64,
step by step (h, the word table) -- is the engine further from float64 than the float32 restatement is?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import util as U
from sert_amd import _capi as C
from oracle import sert_oracle as O
for dims in ({'B': 2500, 'n': 3, 'z': 5, 'Vw': 5, 'Ve': 5000, 'dw': 300, 'de': 128},
             {'B': 64, 'n': 6, 'z': 18, 'Vw': 70000, 'Ve': 2, 'dw': 128, 'de': 128}):
    B, n, z = dims['B'], dims['n'], dims['z']
    steps = 3
    p = U.make_vs_problem(0, B * steps, n, z, dims['Vw'], dims['Ve'], dims['dw'], dims['de'], zipf=True)
    eng = U.vs_engine(p, B, n, z, 0.01)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    o32 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    print(dims)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        neg = p['rng'].randint(0, dims['Ve'], size=(B, z)).astype(np.int64)
        out = []
        for o in (o32, o64):
            l, g, f = o.loss_and_grads(p['X'][sl], p['y'][sl], p['w'][sl], neg)
            o.opt.update(o.params(), g)
            out.append((l, g, f))
        loss = eng.train_batch(s, neg)
        h = eng.get_tensor(C.T_ACT_H, (B, dims['dw']))
        g = eng.get_tensor(C.T_GRAD_RW, (dims['Vw'], dims['dw']))
        Rw = eng.get_tensor(C.T_RW, (dims['Vw'], dims['dw']))
        r = U.rel_err
        print(' step %d loss hip %.7f o32 %.7f o64 %.7f | h: hip-o64 %.2e o32-o64 %.2e | dRw: hip-o64 %.2e o32-o64 %.2e | Rw: hip-o64 %.2e o32-o64 %.2e hip-o32 %.2e' % (
            s, loss, out[0][0], out[1][0], r(h, out[1][2]['h']), r(out[0][2]['h'], out[1][2]['h']),
            r(g, out[1][1][1]), r(out[0][1][1], out[1][1][1]), r(Rw, o64.R_w), r(o32.R_w, o64.R_w), r(Rw, o32.R_w)))
    eng.close()
