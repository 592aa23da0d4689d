"""Every GEMM kernel of the library against float64, through sert_debug_gemm -- the same dispatch (launch_gemm) a training
step uses, so a shape lands on the kernel the step would run it on: the bf16-pipe kernel with exactly split operands
(gemm_x3.h: 128 row tiles or more -- M >= 16384 for N <= 128, M >= 32768 above --, A not transposed, or A^T.B over K >= 4096; SERT_GEMM_FP32=1 sends those shapes to the fp32 MFMA kernels too), the
128x128-tile kernel, the 64x64-tile one, the 128x160-tile one (N just above a multiple of 128) and -- against a variants build (SERT_LIB=.../libsert_variants.so) with
SERT_GEMM_DIRECT_MIN_K=256 / SERT_GEMM_STREAM=1 -- the two kernels of round 4 that feed A straight from global memory into
v_mfma_f32_16x16x4_f32 (variants/gemm_direct.h, variants/gemm_stream.h; measured equal or slower, not in the product)."""
import numpy as np
import pytest

from tests.util import C

pytestmark = pytest.mark.gpu


def _ref(A, B, ta, tb, epi, bias):
    a = A.astype(np.float64).T if ta else A.astype(np.float64)
    b = B.astype(np.float64).T if tb else B.astype(np.float64)
    c = a @ b
    if epi:
        c = c + bias.astype(np.float64)
    if epi == 2:
        c = np.tanh(c)
    return c


@pytest.mark.parametrize('M,N,K,ta,tb,epi', [
    # the streaming projection kernel: h.W + b -> tanh, da.W^T; ragged last strip, fewer strips than waves x 2, many
    (65536, 128, 128, 0, 0, 2), (65536, 128, 128, 0, 1, 0), (8192 + 5, 128, 128, 0, 0, 2), (8192 + 5, 128, 128, 0, 1, 0),
    (20000, 128, 128, 0, 0, 1), (20000, 128, 128, 0, 1, 1), (16384, 128, 128, 0, 0, 0),
    # just outside its shape conditions: the tiled kernels
    (8191, 128, 128, 0, 0, 2), (65536, 128, 112, 0, 0, 2), (4099, 128, 128, 0, 1, 0), (30000, 112, 128, 0, 1, 0),
    # 64x64 tiles, 128x128 tiles, 128x160 tiles, odd sizes (scalar loaders), A^T.B
    (1000, 300, 300, 0, 0, 2), (4096, 1000, 128, 0, 0, 1), (3000, 128, 1000, 0, 1, 0), (333, 77, 45, 0, 0, 1),
    (128, 128, 5000, 1, 0, 0), (300, 301, 2000, 1, 0, 0), (257, 129, 64, 1, 1, 0),
    # long-K shapes (variants build + SERT_GEMM_DIRECT_MIN_K=256: gemm_direct.h): 256-row and 128-row tiles, both B layouts, ragged M, N
    # (last column tile 104 / 44 wide), K with a partial last slab and a partial last 16-k block (1000 = 15 x 64 + 40)
    (65536, 128, 1000, 0, 0, 0), (65536, 128, 1000, 0, 1, 0), (20000, 1000, 256, 0, 1, 1), (44467, 128, 1000, 0, 1, 0),
    (9000, 300, 300, 0, 0, 2), (9001, 300, 300, 0, 1, 0), (4096, 1000, 260, 0, 0, 1), (2049, 256, 4096, 0, 0, 0),
    (4096, 4096, 512, 0, 1, 0),
    # gemm_x3.h (every shape above with 128 row tiles or more goes there too): its three tile shapes (128, 256 and 320
    # columns), one and several column tiles, both B layouts, ragged M / N / K (K = 36: two full steps and a quarter)
    (8192 + 100, 200, 100, 0, 0, 1), (8192 + 100, 200, 100, 0, 1, 1), (10000, 320, 304, 0, 1, 0), (12000, 130, 36, 0, 0, 2),
    (65536, 300, 300, 0, 0, 2), (65536, 300, 300, 0, 1, 0), (8200, 257, 64, 0, 0, 0), (8200, 31, 16, 0, 1, 0),
    (16384 + 7, 128, 128, 0, 0, 2), (32768 + 9, 300, 300, 0, 1, 0), (32768, 257, 64, 0, 0, 1), (16400, 31, 16, 0, 1, 0), (33000, 1000, 128, 0, 0, 1),
    (33000, 1000, 128, 0, 1, 0), (32768, 700, 36, 0, 0, 2),
    # few rows, a wide N, K >= 256: a tile for every CU from the columns (the loglinear logits over 100 000 entities)
    (2304, 20000, 300, 0, 0, 1), (1100, 30000, 256, 0, 1, 0), (1030, 80000, 260, 0, 0, 0),
    # mid-size products in 128 x 128 tiles, any alignment (the loglinear model of a batch of 1024 over 715 experts: N and the
    # leading dimensions no multiple of four, K = 715 with a partial last piece and a partial last step)
    (3500, 715, 300, 0, 0, 1), (3500, 300, 715, 0, 1, 0), (4099, 513, 301, 0, 0, 2), (2050, 1001, 263, 0, 1, 1), (3000, 716, 300, 0, 0, 0),
])
def test_gemm_dispatch_against_float64(hip_lib, M, N, K, ta, tb, epi):
    rng = np.random.RandomState(M + 3 * N + 7 * K + ta + 2 * tb)
    A = rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    B *= np.float32(1.0 / np.sqrt(K))
    bias = rng.uniform(-0.5, 0.5, N).astype(np.float32) if epi else None
    got = C.debug_gemm(A, B, ta=ta, tb=tb, epi=epi, bias=bias)
    ref = _ref(A, B, ta, tb, epi, bias)
    assert np.all(np.isfinite(got))                       # (the output is pre-filled with NaN: every element was written)
    # fp32 accumulation over K terms of magnitude <= 1/sqrt(K): ~1e-6; fast_tanh adds <= 4 ulp
    assert np.abs(got - ref).max() < (2e-6 if epi != 2 else 3e-6) * max(1.0, np.sqrt(K / 128.0))


@pytest.mark.parametrize('M,N,K,splits', [
    (128, 128, 65536, 512), (300, 300, 65536, 113), (128, 128, 8192 + 24, 7), (300, 300, 4096 + 16, 3), (300, 160, 20000, 40),
    (100, 36, 16384, 64), (320, 320, 8192, 8), (301, 299, 9000, 5),
    # several 128 x 128 output tiles per k range (the loglinear dW, the full softmax's dR_e), ragged last tiles
    (128, 1000, 20000, 20), (1000, 128, 16384, 16), (400, 128, 8192, 32), (130, 260, 4096 + 48, 3),
    # a shorter K with an output wide enough to fill the machine by itself (the loglinear dW over 100 000 entities): one k range
    (300, 41000, 1040, 1), (100, 33000, 1536, 2), (257, 36000, 1100, 1),
    # ... or k ranges enough (the loglinear dW of a batch of 1024: 300 x 715 over ~3 500 distinct words in 57 ranges)
    (300, 715, 3500, 57), (128, 715, 2000, 40),
    # outside gemm_x3.h's split-K shapes (K < 4096): the fp32 MFMA kernels
    (128, 128, 2048, 16), (400, 128, 2048, 8),
])
def test_split_k_with_column_sums_against_float64(hip_lib, M, N, K, splits):
    """dW = h^T.da and db = the column sums of da, as a training step computes them: split-K partial slabs with the column
    sums riding along, combined in a fixed order."""
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A = rng.uniform(-1, 1, (K, M)).astype(np.float32)
    B = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
    got, colsum = C.debug_gemm_splitk(A, B, splits)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    assert np.all(np.isfinite(got)) and np.all(np.isfinite(colsum))
    # fp32 accumulation: a chain of K / splits terms of magnitude <= 1 / sqrt(K) per k range (the bound of the test above
    # for that length), then the order-fixed combine of the ranges
    tol = 2e-6 * max(1.0, np.sqrt(K / splits / 128.0), np.sqrt(K / 128.0) / 8)
    assert np.abs(got - ref).max() < tol
    assert np.abs(colsum - B.astype(np.float64).sum(axis=0)).max() < tol


@pytest.mark.parametrize('tb', [0, 1])
def test_non_finite_operands_stay_confined_to_their_rows_and_columns(hip_lib, tb):
    """gemm_x3.h splits an operand as x0 = bf16(x), x1 = bf16(x - x0), ...: an Inf gives Inf - Inf = NaN in the second piece, so
    where the fp32 MFMA path may return +-Inf this one returns NaN -- either way non-finite (a training step raises on a
    non-finite loss, sert/models.py:372-379), and only in the row of A / column of B the value sits in."""
    M, N, K = 16384 + 64, 128, 128
    rng = np.random.RandomState(7)
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = (rng.uniform(-1, 1, (N, K) if tb else (K, N)) / np.sqrt(K)).astype(np.float32)
    A[5, 17] = np.inf
    A[9000, 100] = np.nan
    if tb:
        B[77, 3] = -np.inf
    else:
        B[3, 77] = -np.inf
    got = C.debug_gemm(A, B, ta=0, tb=tb)
    bad = ~np.isfinite(got)
    assert bad[5].all() and bad[9000].all() and bad[:, 77].all()
    bad[5] = bad[9000] = False
    bad[:, 77] = False
    assert not bad.any()


@pytest.mark.parametrize('M,N,K,splits,tb', [
    (2304, 300, 100000, 64, 1), (2304, 300, 50000 + 24, 37, 0), (1100, 128, 65536, 128, 1),
    # a medium K in a few ranges, unaligned (the loglinear dG over 715 experts)
    (3500, 300, 715, 5, 1), (3500, 300, 715, 5, 0),
    # too few tiles x ranges for gemm_x3.h: the fp32 MFMA kernels
    (512, 128, 8192, 8, 1),
])
def test_long_k_in_ranges_against_float64(hip_lib, M, N, K, splits, tb):
    """A.op(B) over a long K cut into k ranges with partial slabs and an order-fixed combine -- the loglinear dG = dZ.W^T over
    a large entity vocabulary (gemm_long_k)."""
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = (rng.uniform(-1, 1, (N, K) if tb else (K, N)) / np.sqrt(K)).astype(np.float32)
    got = C.debug_gemm_longk(A, B, splits, tb=tb)
    ref = A.astype(np.float64) @ (B.astype(np.float64).T if tb else B.astype(np.float64))
    assert np.all(np.isfinite(got))
    assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.sqrt(K / splits / 128.0), np.sqrt(K / 128.0) / 8)


def test_dispatch_on_seeded_random_shapes(hip_lib):
    """Thirty seeded shapes around the dispatch thresholds of gemm_x3.h (row counts about 1024 and 128 tiles, column counts
    about 128 / 256 / 320 and not multiples of four, K about 256 and odd): whichever kernel takes a shape, the result
    is the float64 product to the fp32 accumulation bound."""
    rng = np.random.RandomState(2024)
    for case in range(30):
        tb = int(rng.randint(2))
        epi = int(rng.randint(3)) if not tb else int(rng.choice([0, 1]))
        M = int(rng.choice([1000, 1024, 1500, 3000, 4096, 9000, 16384, 16500, 33000]) + rng.randint(0, 40))
        N = int(rng.choice([7, 100, 127, 128, 129, 200, 256, 257, 300, 320, 321, 715, 1000]))
        K = int(rng.choice([16, 36, 100, 255, 256, 257, 300, 301, 512, 715]))
        A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        B = (rng.uniform(-1, 1, (N, K) if tb else (K, N)) / np.sqrt(K)).astype(np.float32)
        bias = rng.uniform(-0.5, 0.5, N).astype(np.float32) if epi else None
        got = C.debug_gemm(A, B, ta=0, tb=tb, epi=epi, bias=bias)
        ref = _ref(A, B, 0, tb, epi, bias)
        assert np.all(np.isfinite(got)), (case, M, N, K, tb, epi)
        err = np.abs(got - ref).max()
        assert err < (2e-6 if epi != 2 else 3e-6) * max(1.0, np.sqrt(K / 128.0)), (case, M, N, K, tb, epi, err)


def test_k_range_forms_on_seeded_random_shapes(hip_lib):
    """Sixteen seeded shapes of A^T.B (+ column sums) in k ranges and eight of A.op(B) in k ranges, around the thresholds
    of the split-K forms of gemm_x3.h (128 / 320 rows, 160-column tiles, K about 1024 and 4096, ragged last ranges)."""
    rng = np.random.RandomState(77)
    for case in range(16):
        M = int(rng.choice([36, 128, 129, 300, 320, 321, 400]))
        N = int(rng.choice([36, 128, 160, 161, 300, 715, 1000]))
        K = int(rng.choice([1000, 1024, 2033, 4095, 4096, 9000]) + rng.randint(0, 17))
        splits = int(rng.choice([1, 2, 7, 16, 43]))
        A = rng.uniform(-1, 1, (K, M)).astype(np.float32)
        B = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        got, colsum = C.debug_gemm_splitk(A, B, splits)
        tol = 2e-6 * max(1.0, np.sqrt(K / splits / 128.0), np.sqrt(K / 128.0) / 8)
        assert np.abs(got - A.astype(np.float64).T @ B.astype(np.float64)).max() < tol, (case, M, N, K, splits)
        assert np.abs(colsum - B.astype(np.float64).sum(axis=0)).max() < tol, (case, M, N, K, splits)
    for case in range(8):
        tb = int(rng.randint(2))
        M = int(rng.choice([1024, 2300, 3500]) + rng.randint(0, 9))
        N = int(rng.choice([128, 300, 301]))
        K = int(rng.choice([715, 4096, 20000]) + rng.randint(0, 5))
        splits = int(rng.choice([3, 5, 19]))
        A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        B = (rng.uniform(-1, 1, (N, K) if tb else (K, N)) / np.sqrt(K)).astype(np.float32)
        got = C.debug_gemm_longk(A, B, splits, tb=tb)
        ref = A.astype(np.float64) @ (B.astype(np.float64).T if tb else B.astype(np.float64))
        assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.sqrt(K / splits / 128.0), np.sqrt(K / 128.0) / 8), (case, M, N, K, splits, tb)
