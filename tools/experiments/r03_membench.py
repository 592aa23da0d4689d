#!/usr/bin/env python
"""Round 3: what the memory system of THIS box delivers, and where the 630-850 us spread of the C4
word-table optimiser comes from.  Prints one table; run through gpurun."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sert_amd import _capi as C   # noqa: E402

MB = 1 << 20


def rate(nbytes, us):
    return nbytes / (us * 1e-6) / 1e9


def main():
    C.require_gpu()
    print(C.device_info(0))
    print('--- stream copy / read (GB/s of bytes moved: copy counts read + write)')
    for mb in (64, 400, 1200, 2400):
        for blocks in (2048, 4096, 8192):
            us = C.bench_memory(C.MEMBENCH_COPY, mb * MB, blocks=blocks)
            ur = C.bench_memory(C.MEMBENCH_READ, mb * MB, blocks=blocks)
            print('copy %5d MB blocks %5d: %8.1f us  %7.0f GB/s | read %8.1f us %7.0f GB/s'
                  % (mb, blocks, us, rate(2 * mb * MB, us), ur, rate(mb * MB, ur)))
    print('--- window gather (vs_gather_mean, uniformly random rows; GB/s of rows fetched)')
    for row in (512, 1200):
        for tab_mb in (0.5, 2, 8, 33.5, 51.2, 128, 600):
            out_bytes = 65536 * row
            us = C.bench_memory(C.MEMBENCH_GATHER, out_bytes, table_bytes=int(tab_mb * MB), row_bytes=row, window=10)
            print('gather row %4d B table %6.1f MB: %7.1f us  %7.0f GB/s' % (row, tab_mb, us, rate(out_bytes * 10, us)))
    print('--- optimiser stream (adam_l2: 4 arrays read, 3 written; GB/s over 28 B/element)')
    for name, n in (('C2 word table 12.8 M', 12800000), ('C4 word table 150 M', 150000000)):
        nbytes = n * 4
        for rep in range(3):
            for gap in (C.SEPARATE_ALLOCATIONS, 0, 4096, 65536 + 4096, MB + 4096 * 3, 2 * MB + 256):
                for blocks in ((2048, 4096) if n < 1e8 else (4096, 8192)):
                    us = C.bench_memory(C.MEMBENCH_OPTIMIZER, nbytes, gap_bytes=gap, blocks=blocks, iters=10)
                    print('%s rep %d gap %-10s blocks %5d: %8.1f us  %7.0f GB/s'
                          % (name, rep, 'separate' if gap == C.SEPARATE_ALLOCATIONS else gap, blocks, us,
                             rate(7 * nbytes, us)))
            sys.stdout.flush()


if __name__ == '__main__':
    t0 = time.time()
    main()
    print('done in %.1f s' % (time.time() - t0))
