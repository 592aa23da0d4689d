"""What the vendor library (rocBLAS/hipBLASLt through torch.matmul, fp32, no TF32) does on
the GEMM shapes of the hot path -- a yardstick for gemm.h, not part of the product."""
import time
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda'
SHAPES = [
    ('vs fwd  65536x128 . 128x128   ', 65536, 128, 128, False),
    ('ll fwd  81920x128 . 128x1000  ', 81920, 1000, 128, False),
    ('ll dG   81920x1000 . 1000x128 ', 81920, 128, 1000, False),
    ('ll dW   128x81920 . 81920x1000', 128, 1000, 81920, False),
    ('query   10000x128 . (100000x128)^T', 10000, 100000, 128, True),
    ('square  4096^3               ', 4096, 4096, 4096, False),
    # round 3: the shapes of the current steps (loglinear over ~44.5 k distinct words; additive full softmax; C4)
    ('ll fwd  44467x128 . 128x1000  ', 44467, 1000, 128, False),
    ('ll dG   44467x1000 . (128x1000)^T', 44467, 128, 1000, True),
    ('fs log  65536x128 . (1000x128)^T', 65536, 1000, 128, True),
    ('fs dp   65536x1000 . 1000x128 ', 65536, 128, 1000, False),
    ('c4 fwd  65536x300 . 300x300   ', 65536, 300, 300, False),
    ('c4 dh   65536x300 . (300x300)^T', 65536, 300, 300, True),
    ('c4fs    65536x300 . (100000x300)^T', 65536, 100000, 300, True),
]
for name, M, N, K, tb in SHAPES:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    f = (lambda: a @ b.t()) if tb else (lambda: a @ b)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    it = 10
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / it * 1e6
    print('%s %9.1f us %7.1f TFLOP/s' % (name, us, 2.0 * M * N * K / us / 1e6))
