"""CPU: the multithreaded C baseline (oracle/sert_cpu.c, timed by bench.py's cpu_baseline leg)
is the same arithmetic as the numpy oracle -- loss rel 1e-5, parameters rel 1e-4 after 3 steps."""
import numpy as np
import pytest

from oracle import cpu_baseline as CB
from oracle import sert_oracle as O


@pytest.mark.parametrize('dims', [
    dict(B=64, n=5, z=4, Vw=500, Ve=37, dw=32, de=48),
    dict(B=96, n=3, z=7, Vw=200, Ve=11, dw=30, de=70),
    dict(B=256, n=10, z=10, Vw=3000, Ve=100, dw=128, de=128),
    dict(B=2048, n=10, z=3, Vw=400, Ve=50, dw=16, de=16),     # heavy words: summed in pieces
])
def test_c_baseline_matches_numpy_oracle(dims):
    rng = np.random.RandomState(0)
    B, n, z, Vw, Ve, dw, de = (dims[k] for k in ('B', 'n', 'z', 'Vw', 'Ve', 'dw', 'de'))
    Rw, Re = O.glorot_uniform(rng, (Vw, dw)), O.glorot_uniform(rng, (Ve, de))
    W, b = O.glorot_uniform(rng, (dw, de)), (0.1 * rng.randn(de)).astype(np.float32)
    ora = O.VectorSpaceOracle(B, n, z, Rw, Re, W, b, 0.01)
    cpu = CB.VectorSpaceCPU(B, n, z, Rw, Re, W, b, 0.01)
    for _ in range(3):
        X = np.minimum(rng.zipf(1.1, size=(B, n)) - 1, Vw - 1)
        y = rng.randint(0, Ve, B)
        w = rng.uniform(0.5, 2.0, B).astype(np.float32)
        neg = rng.randint(0, Ve, (B, z))
        ref, got = ora.train_step(X, y, w, neg), cpu.train_step(X, y, w, neg)
        assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    p = cpu.params()
    for name, ref in (('R_w', ora.R_w), ('R_e', ora.R_e), ('W', ora.W), ('b', ora.b)):
        assert np.abs(p[name] - ref).max() <= 1e-4 * np.abs(ref).max(), name
    cpu.close()


def test_c_scoring_matches_numpy_oracle():
    rng = np.random.RandomState(1)
    E, P = rng.randn(3000, 48).astype(np.float32), rng.randn(17, 48).astype(np.float32)
    idx = CB.score_topk(E, P, 25)
    for q in range(P.shape[0]):
        order, _ = O.vectorspace_rank(P[q], E, top=25)
        assert np.array_equal(order, idx[q])
