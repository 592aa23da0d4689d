"""Workload for a kernel trace WITHOUT the per-step host round trip: sert_train_batches (one host call, no run-ahead, no spin).
    python r05_inner_batches.py B [entities dim entity_dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from sert_amd import models
B = int(sys.argv[1]); Ve = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128; de = int(sys.argv[4]) if len(sys.argv) > 4 else d
rng = np.random.RandomState(0)
X, y, w = bench.synth_data(rng, 8 * B, 10, 100000, Ve)
m = bench.build_model('vectorspace', models, B, 10, 100000, Ve, d, de, 10, X, y, w, seed=0)
eng = m._engine
for _ in range(3):
    eng.train_batches([i % 8 for i in range(40)])
eng.synchronize()
