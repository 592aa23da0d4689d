#!/usr/bin/env python
"""Round 3: is the 640-840 us spread of the C4 word-table update a property of where its four 600 MB arrays land?
The same model is built, timed (per-kernel HIP events) and destroyed several times in ONE process, and again in
fresh processes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CODE = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
import bench
from sert_amd import models, distributed as dist
B, n, Vw, Ve, d, z = 65536, 10, 500000, 100000, 300, 10
rng = np.random.RandomState(0)
X, y, w = bench.synth_data(rng, 2 * B, n, Vw, Ve)
out = []
for rep in range(%d):
    m = bench.build_model('vectorspace', models, B, n, Vw, Ve, d, d, z, X, y, w, seed=0)
    _, tm, _ = bench.timed_steps(m, dist, 2, 8, 2, timing=True)
    dt, _, _ = bench.timed_steps(m, dist, 2, 20, 5, timing=False)
    out.append((round(tm['optimizer_word_table'], 1), round(tm['optimizer_other'], 1), round(1e3 * dt / 20, 3)))
    del m
print('adam_word_us, adam_entity_us, step_ms:', out)
'''

if __name__ == '__main__':
    for reps in (5, 1, 1, 1):
        r = subprocess.run([sys.executable, '-c', CODE % (ROOT, reps)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        print(r.stdout.decode().strip().splitlines()[-1])
