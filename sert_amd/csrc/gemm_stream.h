// Projection-shaped fp32 GEMMs as STREAMING kernels: a huge M against a small, LDS-resident B (gfx950).
//
//   vectorspace  a  = tanh(h.W + b)     (B, d_w) x (d_w, d_e)     sert/models.py:1055-1061      B stored (K, N)
//                dh = da.W^T            (B, d_e) x (d_w, d_e)^T   (autodiff of the above)        B stored (N, K)
//
// At C2 these are 65536 x 128 x 128: one 128 x 128 output tile with EIGHT k-steps of sixteen.  The tiled kernel
// (gemm.h) loads a tile, multiplies, stores, with two tiles per CU -- 30 us against 13.7 us of MFMA time and 11 us of
// HBM time that could overlap; so did every LDS-staged strip variant of rounds 1-3 (28-35 us).  Here the A operand never
// touches LDS and there is NO barrier in the main loop:
//
//   * v_mfma_f32_16x16x4_f32 takes A as one value per lane: lane l holds A[l % 16][l / 16].  A lane that loads the
//     16 bytes A[row][16 j + 4 q .. + 3] (q = l / 16) straight from global memory owns the A operand of FOUR k-steps:
//     step (j, i) uses component i, i.e. k = 16 j + 4 q + i on the lanes of quarter q -- a permutation of the sixteen
//     k of block j, applied to B as well, so every product a_k b_k is still formed exactly once (only the order of the
//     fp32 additions differs from gemm.h).  A 16-row strip of A is K / 16 global_load_dwordx4 per lane.
//   * B (the 64 KB projection matrix) is copied to LDS once per workgroup and read as ds_read_b128:
//       B stored (K, N): lane (c = l % 16, q) reads B[k][64 g + 4 c .. + 3] -- one read feeds the step of FOUR 16-column
//                        blocks (block b of group g owns the columns 64 g + 4 c + b: a column permutation the epilogue
//                        undoes for free, a lane ends up with four CONSECUTIVE columns and stores them as 16 bytes);
//       B stored (N, K): lane reads B[16 nb + c][16 j + 4 q .. + 3] -- one read feeds four k-steps of one block.
//   * a wave owns whole strips: load (the next strip's loads are in flight under the current strip's MFMAs),
//     K / 4 x N / 16 MFMAs, epilogue.  Eight waves per CU, each at its own point of that cycle, keep the matrix pipe fed
//     where the tiled kernel's waves all wait at the same barrier.
//
// N <= 128, K <= 128, both multiples of 16 (the tables of a d = 128 model); everything else stays on gemm.h.
#pragma once
#include "common.h"
#include "gemm.h"

namespace sert {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct StreamGemmArgs {
    const float* A;      // (M, K) row-major, lda
    const float* B;      // BRC: (K, N) row-major, ldb   else: (N, K) row-major, ldb
    float* C;            // (M, N) row-major, ldc
    const float* bias;   // (N) or null
    int M, lda, ldb, ldc;
    int nstrips;         // cdiv(M, 16)
};

constexpr int kStreamWaves = 8;

template <bool BRC, int EPI, int NB, int KB>
__global__ __launch_bounds__(64 * kStreamWaves, 1) void gemm_stream_f32(const StreamGemmArgs g) {
    constexpr int N = 16 * NB, K = 16 * KB;
    constexpr int LDB = BRC ? N : K + 4;              // (N, K) rows padded: 16 consecutive rows hit 16 distinct bank quads
    constexpr int BROWS = BRC ? K : N;
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [BROWS][LDB]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    // ---- B -> LDS, memory layout = LDS layout: 16-byte pieces, no transposition ----
    {
        constexpr int PIECES_PER_ROW = (BRC ? N : K) / 4;
        for (int p = threadIdx.x; p < BROWS * PIECES_PER_ROW; p += 64 * kStreamWaves) {
            const int r = p / PIECES_PER_ROW, cc = (p - r * PIECES_PER_ROW) * 4;
            *reinterpret_cast<float4*>(&Bs[r * LDB + cc]) = *reinterpret_cast<const float4*>(g.B + (size_t)r * g.ldb + cc);
        }
    }
    // ---- this wave's strips: s = first, first + stride, ... ----
    const int stride = gridDim.x * kStreamWaves;
    int s = blockIdx.x * kStreamWaves + wv;
    float4 a[KB];
    auto load_strip = [&](int strip, float4 (&dst)[KB]) {
        const int row = min(strip * 16 + c, g.M - 1);               // (a ragged last strip re-reads the last row)
        const float* p = g.A + (size_t)row * g.lda + 4 * q;
#pragma unroll
        for (int j = 0; j < KB; ++j) dst[j] = *reinterpret_cast<const float4*>(p + 16 * j);
    };
    if (s < g.nstrips) load_strip(s, a);
    // bias of this lane's columns
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = 0.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = g.bias[BRC ? 64 * (nb >> 2) + 4 * c + (nb & 3) : 16 * nb + c];
    }
    __syncthreads();
    while (s < g.nstrips) {
        const int sn = s + stride;
        float4 an[KB];
        if (sn < g.nstrips) load_strip(sn, an);
        f32x4v acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4v)(0.f);
        if (BRC) {
            // K / 4 steps; the B fragments of step t + 1 are read while the MFMAs of step t issue (one step ahead is
            // enough: LDS latency ~ 64 cycles, a step's eight MFMAs occupy the pipe for 256)
            float4 bcur[NB / 4], bnxt[NB / 4];
#pragma unroll
            for (int gq = 0; gq < NB / 4; ++gq) bcur[gq] = *reinterpret_cast<const float4*>(&Bs[(4 * q) * LDB + 64 * gq + 4 * c]);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 4 * j + i;
                    if (t + 1 < 4 * KB) {
                        const int kn = 16 * ((t + 1) >> 2) + 4 * q + ((t + 1) & 3);
#pragma unroll
                        for (int gq = 0; gq < NB / 4; ++gq) bnxt[gq] = *reinterpret_cast<const float4*>(&Bs[kn * LDB + 64 * gq + 4 * c]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gq = 0; gq < NB / 4; ++gq) {
                        acc[4 * gq + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur[gq].x, acc[4 * gq + 0], 0, 0, 0);
                        acc[4 * gq + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur[gq].y, acc[4 * gq + 1], 0, 0, 0);
                        acc[4 * gq + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur[gq].z, acc[4 * gq + 2], 0, 0, 0);
                        acc[4 * gq + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur[gq].w, acc[4 * gq + 3], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gq = 0; gq < NB / 4; ++gq) bcur[gq] = bnxt[gq];
                }
            }
        } else {
            // K / 16 blocks of NB reads; block j + 1's fragments are read while block j's 4 NB MFMAs issue
            float4 bcur[NB], bnxt[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bcur[nb] = *reinterpret_cast<const float4*>(&Bs[(16 * nb + c) * LDB + 4 * q]);
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
                if (j + 1 < KB) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bnxt[nb] = *reinterpret_cast<const float4*>(&Bs[(16 * nb + c) * LDB + 16 * (j + 1) + 4 * q]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bcur[nb].x, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bcur[nb].y, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bcur[nb].z, acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bcur[nb].w, acc[nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bcur[nb] = bnxt[nb];
            }
        }
        // ---- epilogue: D[row = 4 q + r][col of (block, lane c)] = acc[block][r] ----
        const int row0 = s * 16 + 4 * q;
        if (BRC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                if (row < g.M) {
#pragma unroll
                    for (int gq = 0; gq < NB / 4; ++gq) {
                        float4 v = make_float4(acc[4 * gq + 0][r], acc[4 * gq + 1][r], acc[4 * gq + 2][r], acc[4 * gq + 3][r]);
                        if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
                            v.x += bv[4 * gq + 0]; v.y += bv[4 * gq + 1]; v.z += bv[4 * gq + 2]; v.w += bv[4 * gq + 3];
                        }
                        if (EPI == EPI_BIAS_TANH) { v.x = fast_tanh(v.x); v.y = fast_tanh(v.y); v.z = fast_tanh(v.z); v.w = fast_tanh(v.w); }
                        *reinterpret_cast<float4*>(g.C + (size_t)row * g.ldc + 64 * gq + 4 * c) = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                if (row < g.M) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float v = acc[nb][r];
                        if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) v += bv[nb];
                        if (EPI == EPI_BIAS_TANH) v = fast_tanh(v);
                        g.C[(size_t)row * g.ldc + 16 * nb + c] = v;
                    }
                }
            }
        }
        if (sn >= g.nstrips) break;
#pragma unroll
        for (int j = 0; j < KB; ++j) a[j] = an[j];
        s = sn;
    }
}

// true if the launch was taken (see the shape conditions at the top)
template <bool TB, int EPI>
inline bool launch_gemm_stream(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                               int lda, int ldb, int ldc) {
    if (EPI != EPI_STORE && EPI != EPI_BIAS && EPI != EPI_BIAS_TANH) return false;
    if (N != 128 || K != 128 || M < 8192) return false;                      // (the instantiated shape: d_w = d_e = 128)
    if (lda % 4 || ldb % 4 || ldc % 4 || ((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16) return false;
    StreamGemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.nstrips = cdiv(M, 16);
    constexpr bool BRC = !TB;
    constexpr int E = (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) ? EPI : EPI_STORE;
    const size_t lds = (size_t)(BRC ? 128 * 128 : 128 * 132) * sizeof(float);
    auto kern = gemm_stream_f32<BRC, E, 8, 8>;
    static const bool attr_ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!attr_ok) return false;
    const int grid = std::min(256, cdiv(g.nstrips, kStreamWaves));
    SERT_LAUNCH(kern, dim3(grid), dim3(64 * kStreamWaves), lds, s, g);
    return true;
}

}  // namespace sert
