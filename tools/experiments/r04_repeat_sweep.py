"""Run-to-run repeatability of every benchmarked configuration through the Python surface, as the epoch loop drives it (hints
for the following batch, loss read every step): the same 6 steps twice in fresh models, every parameter tensor and the
optimiser state compared bit for bit.  (Two schedule races of round 4 were found by exactly this kind of re-run.)"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sert_amd import models  # noqa: E402

CASES = [
    ('C2 vectorspace', 'vectorspace', dict(B=65536, n=10, Vw=100000, Ve=1000, dw=128, de=128)),
    ('C2 dims loglinear 65536', 'loglinear', dict(B=65536, n=10, Vw=100000, Ve=1000, dw=128, de=128)),
    ('C2 dims full softmax', 'vectorspace_softmax', dict(B=65536, n=10, Vw=100000, Ve=1000, dw=128, de=128)),
    ('C4 vectorspace', 'vectorspace', dict(B=65536, n=10, Vw=500000, Ve=100000, dw=300, de=300)),
    ('C4 loglinear 1024', 'loglinear', dict(B=1024, n=10, Vw=500000, Ve=100000, dw=300, de=300)),
    ('product search', 'vectorspace', dict(B=4096, n=10, Vw=100000, Ve=32768, dw=300, de=128)),
    ('W3C loglinear', 'loglinear', dict(B=1024, n=8, Vw=100000, Ve=715, dw=300, de=300)),
    ('C2 dims vectorspace 16384', 'vectorspace', dict(B=16384, n=10, Vw=100000, Ve=1000, dw=128, de=128)),
]


def run(kind, c):
    rng = np.random.RandomState(0)
    X, y, w = bench.synth_data(rng, 3 * c['B'], c['n'], c['Vw'], c['Ve'])
    m = bench.build_model(kind, models, c['B'], c['n'], c['Vw'], c['Ve'], c['dw'], c['de'], 10, X, y, w, seed=0)
    eng = m._engine
    losses = []
    for s in range(6):
        eng.hint_next_batch((s + 1) % 3 if s < 5 else None)
        losses.append(float(m.train_fn(s % 3)))
    eng.hint_next_batch(None)
    out = {'losses': losses}
    st = m.get_optimizer_state() if hasattr(m, 'get_optimizer_state') else {}
    reps = m.get_representations()
    reps = reps if isinstance(reps, (list, tuple)) else [reps]
    for i, r in enumerate(reps):
        out['rep%d' % i] = zlib.crc32(np.ascontiguousarray(r).tobytes())
    for k, v in (st.items() if isinstance(st, dict) else enumerate(st)):
        if isinstance(v, np.ndarray):
            out['state_%s' % k] = zlib.crc32(np.ascontiguousarray(v).tobytes())
    del m
    return out


for name, kind, c in CASES:
    a, b = run(kind, c), run(kind, c)
    same = a == b
    digest = zlib.crc32(repr(sorted((k, v) for k, v in a.items())).encode())
    print('%-28s %s   losses %s   digest %08x' % (name, 'bit-identical' if same else 'DIFFERENT: %s' % [k for k in a if a[k] != b.get(k)],
                                                   ['%.6f' % x for x in a['losses'][-2:]], digest))
