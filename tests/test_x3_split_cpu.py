"""The arithmetic behind csrc/gemm_x3.h, restated in NumPy (no GPU): a float32 splits EXACTLY into three bfloat16 pieces
(round to nearest even, as v_cvt_pk_bf16_f32 does), and the six products a_p b_q with p + q <= 2 -- each exact in the
bf16 MFMA, accumulated here in float64 -- leave an error two orders of magnitude below that of an fp32 GEMM.  The kernel
itself is pinned against float64 on the GPU (tests/test_gpu_gemm.py); this file pins the claim the kernel rests on."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_rne(x)
    r1 = x - x0                      # exact in float32
    x1 = bf16_rne(r1)
    r2 = r1 - x1                     # exact in float32
    x2 = bf16_rne(r2)
    return x0, x1, x2


def test_three_bf16_pieces_add_up_to_the_float32_bit_for_bit():
    rng = np.random.RandomState(0)
    samples = [rng.standard_normal(200000).astype(np.float32),
               (rng.standard_normal(200000) * 1e-3).astype(np.float32),
               rng.uniform(-1, 1, 200000).astype(np.float32) * np.float32(2.0) ** rng.randint(-60, 60, 200000).astype(np.float32),
               np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0000002, 0.1, 1e-30, -1e30, 255.99998,
                         1.9999999, 2.0 ** -100], dtype=np.float32)]
    for x in samples:
        x0, x1, x2 = split3(x)
        # every piece is a bfloat16 (its low 16 bits are zero) ...
        for piece in (x0, x1, x2):
            assert not np.any(piece.view(np.uint32) & 0xffff)
        # ... and the three add up to x exactly
        total = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
        assert np.array_equal(total, x.astype(np.float64))


def test_six_products_are_closer_to_float64_than_an_fp32_gemm():
    rng = np.random.RandomState(1)
    M, K, N = 256, 300, 300
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    a, b = split3(A), split3(B)
    six = sum(a[p].astype(np.float64) @ b[q].astype(np.float64) for p in range(3) for q in range(3) if p + q <= 2)
    nine = sum(a[p].astype(np.float64) @ b[q].astype(np.float64) for p in range(3) for q in range(3))
    err6 = (np.abs(six - ref) / scale).max()
    err9 = (np.abs(nine - ref) / scale).max()
    err32 = (np.abs((A @ B).astype(np.float64) - ref) / scale).max()
    assert err9 < 1e-15                       # nine products: the exact product of the exact splits
    assert err6 < 3 * 2.0 ** -24 * 0.2        # the three dropped ones: far below 3 . 2^-24 |a||b| (they rarely align)
    assert err6 < err32 / 20                  # ... and far below what the fp32 accumulation alone costs
    # three products (the "bf16 x 3" of some libraries) would NOT do: 1e-6, an order above fp32
    three = sum(a[p].astype(np.float64) @ b[q].astype(np.float64) for p, q in ((0, 0), (0, 1), (1, 0)))
    assert (np.abs(three - ref) / scale).max() > 2 * err32
