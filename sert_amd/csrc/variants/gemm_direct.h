// fp32 GEMM with the A operand fed to the matrix cores STRAIGHT FROM GLOBAL MEMORY (gfx950).
//
//   C (M, N) = epi( A (M, K) . op(B) ),  A row-major (contiguous along k);  B stored (K, N) ["rc"] or (N, K) ["kc"]
//
// The long-K GEMMs of the path -- the full-softmax variants' dp = dZ.R_e (65536 x 128 x 1000) and logits, the loglinear
// forward and dG = dZ.W^T, the d = 300 projections -- spend most of gemm.h's time outside the MFMAs: its 128 x 128 x 16
// tiles stage BOTH operands through LDS with a barrier every 16 k, a k-contiguous operand is transposed on the way in
// with four ds_write_b32 per 16 bytes, and every MFMA operand is a ds_read_b32 (round 3: 2.35 VALU + 0.69 LDS
// instructions per MFMA, 73 % of the wave cycles waiting; the vendor library's kernel: 0.14 + 0.38).  Here:
//
//   * v_mfma_f32_16x16x4_f32, A as one value per lane (lane l: row l % 16, k-quarter q = l / 16).  A lane that loads the
//     16 bytes A[row][16 j + 4 q .. + 3] owns the A operand of the FOUR k-steps of 16-k block j: step i uses component
//     i, i.e. k = 16 j + 4 q + i -- a permutation of the block's sixteen k that B follows.  A never touches LDS: a wave
//     owns 16 RB rows of the tile and prefetches a whole 64-k slab of them (4 RB loads) one slab ahead.
//   * B goes through LDS in 64-k slabs, LDS layout = memory layout (16-byte stores, no transposition), read as
//     ds_read_b128:  rc: B[k][64 g + 4 c .. + 3] feeds the step of FOUR 16-column blocks (block b of group g owns the
//     columns 64 g + 4 c + b -- the epilogue undoes the permutation for free: a lane ends up with four CONSECUTIVE
//     columns and stores 16 bytes);  kc: B[16 nb + c][16 j + 4 q .. + 3] feeds four k-steps of one block.
//   * one barrier per 64 k (gemm.h: per 16), 0.125-0.25 LDS reads per MFMA, no VALU address work in the loop.
//
// Tile 16 RB WAVES x 128 (8 waves: 128 or 256 rows).  Every product a_k b_k is formed exactly once; the order of the fp32
// additions differs from gemm.h (k permuted inside blocks of sixteen).  K % 4 == 0, 16-byte aligned operands.
//
// MEASURED, NOT IN THE PRODUCT (round 4; profiles/r04_experiments.txt, tools/experiments/r04_gemm_direct.sh): exact against
// float64 on every shape (tests/test_gpu_gemm.py on a variants build), equal to gemm.h with B stored (K, N) -- the
// full-softmax dp 179.8 against 180.1 us, 4096^3 111 against 108 TF -- and SLOWER with B stored (N, K): loglinear dG 160.7
// against 132.2 us; d = 300 pays 384 columns for 300.  Compiled with -DSERT_VARIANTS; SERT_GEMM_DIRECT_MIN_K=k routes
// operands with K >= k to it.
#pragma once
#include "../common.h"
#include "../gemm.h"

namespace sert {

typedef float f32x4d __attribute__((ext_vector_type(4)));

struct DirectGemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int tiles_n;
};

constexpr int DG_WAVES = 8, DG_BN = 128, DG_KS = 64;

template <bool BRC, int EPI, int RB>
__global__ __launch_bounds__(64 * DG_WAVES, 1) void gemm_f32_direct(const DirectGemmArgs g) {
    constexpr int NB = DG_BN / 16;                       // 16-column blocks per tile
    constexpr int KB = DG_KS / 16;                       // 16-k blocks per slab
    constexpr int LDB = BRC ? DG_BN : DG_KS + 4;         // kc rows padded: 16 consecutive rows hit 16 distinct bank quads
    constexpr int BROWS = BRC ? DG_KS : DG_BN;
    constexpr int SLAB = BROWS * LDB;                    // floats per LDS buffer
    constexpr int BP = (DG_KS * DG_BN / 4) / (64 * DG_WAVES);   // 16-byte pieces of a B slab per thread (4)
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [2][SLAB]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
    const int m0 = tm * (16 * RB * DG_WAVES), n0 = tn * DG_BN;
    const int nslabs = (g.K + DG_KS - 1) / DG_KS;

    // ---- A: this lane's rows (clamped: a ragged tile re-reads the last row, its stores are guarded) ----
    const float* arow[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = min(m0 + (wv * RB + rb) * 16 + c, g.M - 1);
        arow[rb] = g.A + (size_t)row * g.lda + 4 * q;
    }
    // Loads are issued from CLAMPED (always valid) addresses and not touched until they are consumed: the zeroing of the
    // pieces beyond K / N happens where the staged values are stored to LDS / promoted to the current slab -- a select
    // right behind the load would make the wave wait for it on the spot and turn the prefetch into a synchronous load.
    auto load_a = [&](int slab, float4 (&dst)[KB][RB]) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const bool ok = slab * DG_KS + 16 * j + 4 * q < g.K;          // (K % 4 == 0: a piece is wholly inside or outside)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dst[j][rb] = *reinterpret_cast<const float4*>(arow[rb] + (ok ? slab * DG_KS + 16 * j : 0));
        }
    };
    auto promote_a = [&](int slab, float4 (&dst)[KB][RB], const float4 (&src)[KB][RB]) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const bool ok = slab * DG_KS + 16 * j + 4 * q < g.K;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dst[j][rb] = ok ? src[j][rb] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // ---- B slab: global -> registers -> LDS ----
    auto b_piece = [&](int slab, int u, int& k, int& n) {
        const int p = threadIdx.x + 64 * DG_WAVES * u;
        if (BRC) { k = slab * DG_KS + p / (DG_BN / 4); n = n0 + 4 * (p % (DG_BN / 4)); }
        else     { n = n0 + p / (DG_KS / 4); k = slab * DG_KS + 4 * (p % (DG_KS / 4)); }
        return k < g.K && n < g.N;                        // (rc: N % 4 == 0; kc: K % 4 == 0 -- pieces never straddle)
    };
    auto load_b = [&](int slab, float4 (&dst)[BP]) {
#pragma unroll
        for (int u = 0; u < BP; ++u) {
            int k, n;
            const bool ok = b_piece(slab, u, k, n);
            const float* src = BRC ? g.B + (size_t)(ok ? k : 0) * g.ldb + (ok ? n : 0) : g.B + (size_t)(ok ? n : 0) * g.ldb + (ok ? k : 0);
            dst[u] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto store_b = [&](int slab, int buf, const float4 (&src)[BP]) {
#pragma unroll
        for (int u = 0; u < BP; ++u) {
            const int p = threadIdx.x + 64 * DG_WAVES * u;
            const int r = BRC ? p / (DG_BN / 4) : p / (DG_KS / 4);
            const int cc = BRC ? 4 * (p % (DG_BN / 4)) : 4 * (p % (DG_KS / 4));
            int k, n;
            const bool ok = b_piece(slab, u, k, n);
            *reinterpret_cast<float4*>(&Bs[buf * SLAB + r * LDB + cc]) = ok ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float4 a_cur[KB][RB], a_nxt[KB][RB], b_stage[BP];
    load_a(0, a_nxt);
    load_b(0, b_stage);
    // bias of this lane's columns
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = 0.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = n0 + (BRC ? 64 * (nb >> 2) + 4 * c + (nb & 3) : 16 * nb + c);
            bv[nb] = col < g.N ? g.bias[col] : 0.f;
        }
    }
    f32x4d acc[RB][NB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4d)(0.f);
    store_b(0, 0, b_stage);
    promote_a(0, a_cur, a_nxt);
    __syncthreads();

    for (int s = 0; s < nslabs; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nslabs;
        if (more) { load_a(s + 1, a_nxt); load_b(s + 1, b_stage); }
        const float* Bb = Bs + buf * SLAB;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (BRC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 * j + 4 * q + i;
                    float4 b[NB / 4];
#pragma unroll
                    for (int gq = 0; gq < NB / 4; ++gq) b[gq] = *reinterpret_cast<const float4*>(&Bb[k * LDB + 64 * gq + 4 * c]);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float av = i == 0 ? a_cur[j][rb].x : i == 1 ? a_cur[j][rb].y : i == 2 ? a_cur[j][rb].z : a_cur[j][rb].w;
#pragma unroll
                        for (int gq = 0; gq < NB / 4; ++gq) {
                            acc[rb][4 * gq + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[gq].x, acc[rb][4 * gq + 0], 0, 0, 0);
                            acc[rb][4 * gq + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[gq].y, acc[rb][4 * gq + 1], 0, 0, 0);
                            acc[rb][4 * gq + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[gq].z, acc[rb][4 * gq + 2], 0, 0, 0);
                            acc[rb][4 * gq + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[gq].w, acc[rb][4 * gq + 3], 0, 0, 0);
                        }
                    }
                }
            } else {
                float4 b[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const float4*>(&Bb[(16 * nb + c) * LDB + 16 * j + 4 * q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const float av = i == 0 ? a_cur[j][rb].x : i == 1 ? a_cur[j][rb].y : i == 2 ? a_cur[j][rb].z : a_cur[j][rb].w;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float bvv = i == 0 ? b[nb].x : i == 1 ? b[nb].y : i == 2 ? b[nb].z : b[nb].w;
                            acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv, acc[rb][nb], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (more) {
            store_b(s + 1, buf ^ 1, b_stage);
            promote_a(s + 1, a_cur, a_nxt);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = 4 q + r][column of (block, lane c)] = acc[rb][block][r] ----
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + (wv * RB + rb) * 16 + 4 * q + r;
            if (row >= g.M) continue;
            float* crow = g.C + (size_t)row * g.ldc;
            if (BRC) {
#pragma unroll
                for (int gq = 0; gq < NB / 4; ++gq) {
                    const int col = n0 + 64 * gq + 4 * c;
                    if (col >= g.N) continue;                       // (N % 4 == 0)
                    float4 v = make_float4(acc[rb][4 * gq + 0][r], acc[rb][4 * gq + 1][r], acc[rb][4 * gq + 2][r], acc[rb][4 * gq + 3][r]);
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) { v.x += bv[4 * gq + 0]; v.y += bv[4 * gq + 1]; v.z += bv[4 * gq + 2]; v.w += bv[4 * gq + 3]; }
                    if (EPI == EPI_BIAS_TANH) { v.x = fast_tanh(v.x); v.y = fast_tanh(v.y); v.z = fast_tanh(v.z); v.w = fast_tanh(v.w); }
                    *reinterpret_cast<float4*>(crow + col) = v;
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int col = n0 + 16 * nb + c;
                    if (col >= g.N) continue;
                    float v = acc[rb][nb][r];
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) v += bv[nb];
                    if (EPI == EPI_BIAS_TANH) v = fast_tanh(v);
                    crow[col] = v;
                }
            }
        }
    }
}

// true if the launch was taken
template <bool TB, int EPI>
inline bool launch_gemm_direct(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                               int lda, int ldb, int ldc) {
    if (EPI != EPI_STORE && EPI != EPI_BIAS && EPI != EPI_BIAS_TANH) return false;
    constexpr bool BRC = !TB;
    if (K % 4 || lda % 4 || ldb % 4 || ((uintptr_t)A | (uintptr_t)B) % 16) return false;
    if (BRC && (N % 4 || ldc % 4 || ((uintptr_t)C) % 16)) return false;
    DirectGemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.tiles_n = cdiv(N, DG_BN);
    constexpr int E = (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) ? EPI : EPI_STORE;
    const size_t lds = (size_t)2 * (BRC ? DG_KS * DG_BN : DG_BN * (DG_KS + 4)) * sizeof(float);
    // 256-row tiles when they still give every CU a tile, 128-row tiles otherwise
    const bool big = (long long)cdiv(M, 256) * g.tiles_n >= 256;
    if (big) {
        auto kern = gemm_f32_direct<BRC, E, 2>;
        static const bool ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        if (!ok) return false;
        SERT_LAUNCH(kern, dim3(cdiv(M, 256) * g.tiles_n), dim3(64 * DG_WAVES), lds, s, g);
    } else {
        auto kern = gemm_f32_direct<BRC, E, 1>;
        static const bool ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        if (!ok) return false;
        SERT_LAUNCH(kern, dim3(cdiv(M, 128) * g.tiles_n), dim3(64 * DG_WAVES), lds, s, g);
    }
    return true;
}

}  // namespace sert
