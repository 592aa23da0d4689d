#!/usr/bin/env python
"""Soak run on the GPU box: many steps of each model through the shipped schedules (run-ahead on), twice in
fresh processes -- every loss finite, and losses + CRCs of all tensors identical between the two runs (the
fence-free tail, the deferred entity-table update and the side-stream schedules leave no room for a race
that a parity test of three steps would not see).   python tools/soak.py [steps_scale]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, json, zlib
sys.path.insert(0, %(root)r)
import numpy as np
import bench
from sert_amd import models
kind, Vw, Ve, d, B, steps = %(kind)r, %(Vw)d, %(Ve)d, %(d)d, %(B)d, %(steps)d
rng = np.random.RandomState(3)
nb = 4
X, y, w = bench.synth_data(rng, nb * B, 10, Vw, Ve)
m = bench.build_model(kind, models, B, 10, Vw, Ve, d, d, 10, X, y, w, seed=3)
eng = m._engine
losses = []
for s in range(steps):
    eng.hint_next_batch((s + 1) %% nb if s + 1 < steps else None)
    losses.append(float(m.train_fn(s %% nb)))
    if s %% 97 == 0:
        losses.append(float(m.test_fn(s %% nb)))      # (an evaluation between two steps: reads every table)
from sert_amd import _capi as C
out = {'finite': bool(np.isfinite(losses).all()), 'first': losses[0], 'last': losses[-1],
       'crc_losses': zlib.crc32(np.asarray(losses, np.float32).tobytes())}
for name, which in (('Rw', C.T_RW), ('W', C.T_W), ('b', C.T_B)) + ((('Re', C.T_RE),) if kind != 'loglinear' else ()):
    out['crc_' + name] = zlib.crc32(eng.get_tensor(which).tobytes())
print('RESULT ' + json.dumps(out))
'''


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    cases = [('vectorspace', 100000, 1000, 128, 65536, int(3000 * scale)),
             ('vectorspace', 500000, 100000, 300, 65536, int(400 * scale)),
             ('loglinear', 100000, 1000, 128, 65536, int(600 * scale)),
             ('vectorspace_softmax', 100000, 1000, 128, 65536, int(500 * scale)),
             ('vectorspace', 20000, 40000, 128, 4096, int(3000 * scale))]      # sorted entity chain + big R_e at a small batch
    bad = 0
    for kind, Vw, Ve, d, B, steps in cases:
        outs = []
        for _ in range(2):
            code = WORKER % dict(root=ROOT, kind=kind, Vw=Vw, Ve=Ve, d=d, B=B, steps=steps)
            r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith('RESULT ')]
            outs.append(json.loads(lines[-1][7:]) if lines else {'error': r.stderr.decode()[-400:]})
        ok = outs[0] == outs[1] and outs[0].get('finite')
        bad += 0 if ok else 1
        print('%-20s V_w=%d V_e=%d d=%d B=%d steps=%d: %s  %s' % (kind, Vw, Ve, d, B, steps, 'identical, finite' if ok else 'MISMATCH', outs[0] if ok else outs))
    print('soak failures:', bad)
    return bad


if __name__ == '__main__':
    sys.exit(main())
