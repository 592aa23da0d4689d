"""How long does a plain float4 copy of the C2 projection's bytes take?  (33.5 MB in, 33.5 MB out: the memory-side floor of
the three 65536 x 128 x 128 GEMMs of the vectorspace step.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sert_amd import _capi as C
MB = 1 << 20
for mb in (8, 16, 32, 64, 128, 512, 1200):
    for blocks in (1024, 2048, 4096):
        us = C.bench_memory(C.MEMBENCH_COPY, mb * MB, blocks=blocks, iters=50)
        usr = C.bench_memory(C.MEMBENCH_READ, mb * MB, blocks=blocks, iters=50)
        print('copy %5d MB -> %5d MB, %4d blocks: %7.2f us = %5.2f TB/s (read + write) | read only %7.2f us = %5.2f TB/s' % (
            mb, mb, blocks, us, 2 * mb * MB / us / 1e6, usr, mb * MB / usr / 1e6))
