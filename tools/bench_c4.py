"""BASELINE.json configs[3] (V_w=500k, V_e=100k, d=300): per-kernel table of one
training step for the three model kinds.  Not the bench line -- a profiling aid.

    python tools/bench_c4.py [--kinds vectorspace,vectorspace_softmax,loglinear]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kinds', default='vectorspace,vectorspace_softmax,loglinear')
    ap.add_argument('--vocab', type=int, default=500000)
    ap.add_argument('--entities', type=int, default=100000)
    ap.add_argument('--dim', type=int, default=300)
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=None, help='override the per-kind default batch')
    args = ap.parse_args()
    from sert_amd import distributed as dist, models
    batch = {'vectorspace': 65536, 'vectorspace_softmax': 8192, 'loglinear': 1024}
    out = {}
    for kind in args.kinds.split(','):
        B = args.batch or batch[kind]
        rng = np.random.RandomState(0)
        X, y, w = bench.synth_data(rng, 2 * B, args.window, args.vocab, args.entities)
        m = bench.build_model(kind, models, B, args.window, args.vocab, args.entities, args.dim,
                              args.dim, 10, X, y, w, seed=0)
        dt, _, loss = bench.timed_steps(m, dist, 2, args.steps, 2, timing=False)
        _, tm, _ = bench.timed_steps(m, dist, 2, args.steps, 1, timing=True)
        out[kind] = {'batch': B, 'ms_per_step': 1000 * dt / args.steps,
                     'pairs_per_s': args.steps * B / dt, 'loss': loss,
                     'kernels_us': {k: round(v, 1) for k, v in tm.items() if v > 0}}
        del m
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
