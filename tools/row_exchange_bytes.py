#!/usr/bin/env python
"""Counted bytes of the data-parallel word-table exchange at the bench's workload, for any number of
ranks, WITHOUT a GPU: the per-rank touched bitmaps of bench.py's synthetic batches go through the
library's own list builder (sert_debug_row_lists = what sert_upload_dataset runs on the gathered
bitmaps).  Prints, per world size, the rows a rank fetches / serves per batch and the bytes it sends +
receives per step by rows against ZeRO-1 (reduce-scatter + all-gather of the padded table).

    python tools/row_exchange_bytes.py [--vocab 100000 --dim 128 --batch 65536 --window 10 --worlds 2,4,8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from sert_amd import _capi, distributed   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vocab', type=int, default=100000)
    ap.add_argument('--entities', type=int, default=1000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--batch', type=int, default=65536, help='per-rank batch (weak scaling, as bench.py)')
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--worlds', default='2,4,8')
    ap.add_argument('--num-batches', type=int, default=2)
    a = ap.parse_args()
    for world in [int(x) for x in a.worlds.split(',')]:
        Bg = a.batch * world
        rng = np.random.RandomState(0)
        X, _, _ = bench.synth_data(rng, a.num_batches * Bg, a.window, a.vocab, a.entities)
        bw = (((a.vocab + 31) // 32) + 3) // 4 * 4
        bits = np.zeros((world, a.num_batches, bw), dtype=np.uint32)
        for r in range(world):
            rows = distributed.shard_rows(len(X), Bg, r, world)
            Xr = X[rows].reshape(a.num_batches, -1)
            for b in range(a.num_batches):
                w = np.unique(Xr[b]).astype(np.int64)
                np.bitwise_or.at(bits[r, b], w >> 5, (np.uint32(1) << (w & 31).astype(np.uint32)))
        R = (-(-a.vocab // world) + 15) // 16 * 16
        fetch, serve, union = [], [], []
        for r in range(world):
            for b in range(a.num_batches):
                L = _capi.debug_row_lists(bits, r, R, a.vocab, b)
                fetch.append(len(L['fetch_rows'])); serve.append(len(L['serve_rows'])); union.append(len(L['union_rows']))
        rowb = a.dim * 4
        by_rows = 2 * (np.mean(fetch) + np.mean(serve)) * rowb
        worst = 2 * (np.max(fetch) + np.max(serve)) * rowb
        zero1 = 2 * 2 * (world - 1) / world * R * world * rowb
        touched = np.mean([np.unpackbits(bits[r, b].view(np.uint8)).sum() for r in range(world) for b in range(a.num_batches)])
        print('N=%d  rows touched per rank-batch %.0f of %d  fetched %.0f  served %.0f (max %d / %d)  owned rows updated with a '
              'gradient %.0f of %d | bytes sent+received per rank and step: by rows %.1f MB (worst rank %.1f MB), ZeRO-1 %.1f MB, ratio %.2f'
              % (world, touched, a.vocab, np.mean(fetch), np.mean(serve), np.max(fetch), np.max(serve), np.mean(union), R,
                 by_rows / 1e6, worst / 1e6, zero1 / 1e6, zero1 / by_rows))


if __name__ == '__main__':
    main()
