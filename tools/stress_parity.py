"""Randomized parity stress (not part of the suite; run on the GPU box): random shapes through the
vectorspace and loglinear step tests."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fails = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    dims = dict(B=int(rng.choice([1, 7, 33, 64, 129, 300, 1000, 2500, 4096])), n=int(rng.randint(1, 13)),
                z=int(rng.randint(1, 21)), Vw=int(rng.choice([5, 60, 300, 5000, 70000])),
                Ve=int(rng.choice([2, 7, 100, 1000, 5000, 40000])),
                dw=int(rng.choice([4, 10, 16, 30, 64, 128, 300])), de=int(rng.choice([4, 12, 36, 64, 128, 300])))
    if dims['B'] * dims['n'] * dims['dw'] > 3e6 or dims['Ve'] * dims['de'] > 6e6:
        continue
    try:
        T.test_vectorspace_steps(None, dims, 'default', None)
        print('ok  vs', dims, flush=True)
    except Exception as e:
        fails += 1
        print('FAIL vs', dims, repr(e)[:200], flush=True)
    lld = dict(B=int(rng.choice([1, 5, 64, 130, 512])), n=int(rng.randint(1, 9)), Vw=int(rng.choice([6, 300, 5000])),
               Ve=int(rng.choice([4, 17, 100, 1000, 3000])), d=int(rng.choice([4, 10, 32, 64, 128])))
    try:
        T.test_loglinear_steps(None, lld, str(rng.choice(['int', 'csr'])))
        print('ok  ll', lld, flush=True)
    except Exception as e:
        fails += 1
        print('FAIL ll', lld, repr(e)[:200], flush=True)
print('failures:', fails)

# scoring: random (V_e, d_e, Q, k) through the oracle check of the fused / materialising paths
for it in range(12):
    V = int(rng.choice([300, 5000, 33000, 40000, 70001, 131073]))
    d = int(rng.choice([4, 8, 20, 32, 64, 100, 128, 256]))
    Q = int(rng.choice([1, 3, 40, 129, 300]))
    k = int(min(V, rng.choice([1, 5, 100, 129, 500, 1000])))
    if V * d > 1.2e7:
        continue
    r2 = np.random.RandomState(rng.randint(1 << 30))
    E = r2.randn(V, d).astype(np.float32)
    if rng.rand() < 0.3:      # near-duplicate cluster
        c = r2.choice(V, min(V, 800), replace=False)
        E[c] = E[c[0]] + 1e-3 * r2.randn(c.size, d).astype(np.float32)
    Pj = np.tanh(r2.randn(Q, d)).astype(np.float32)
    try:
        idx, val = T.C.score_topk(E, Pj, k)
        T._check_topk_against_oracle(E, Pj, idx, val, k)
        print('ok  score', dict(V=V, d=d, Q=Q, k=k), flush=True)
    except Exception as e:
        fails += 1
        print('FAIL score', dict(V=V, d=d, Q=Q, k=k), repr(e)[:200], flush=True)
print('failures incl. scoring:', fails)
