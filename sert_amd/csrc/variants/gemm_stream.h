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
//
// MEASURED EQUAL, NOT IN THE PRODUCT (round 4; profiles/r04_experiments.txt): 28.8-29.7 us (projection + tanh) and
// 27.6-28.8 us (dh) against 30.5 / 29.5 us for the tiled kernels and 26.7 us for the vendor library's plain GEMM.
// Knock-outs (SERT_STREAM_KO bits; timing only): launch + B copy + first loads 5.7 us; + the MFMA loop 21.8 (the loop
// itself: 16.1 us = 85 % of the fp32 MFMA peak); + the A loads 22.3 (hidden); + bias / tanh / store INSTRUCTIONS with
// every store landing in one 512 KB region 27.6 (dh: 24.8); + the real stores 28.8 (write-through sc1 stores: 1.8 us
// better than plain ones).  So the A operand needs neither LDS nor barriers, and the loop runs near the pipe's
// rate -- what is left is a 5.7 us prologue and a 5-7 us epilogue that does not hide under the partner wave's MFMAs
// (starting the second wave of each SIMD half a unit late, 4 or 2 waves per workgroup: no better).
// Compiled with -DSERT_VARIANTS only; SERT_GEMM_STREAM=1 routes the shape to it.
#pragma once
#include "../common.h"
#include "../gemm.h"

namespace sert {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// 16-byte WRITE-THROUGH store (sc1): the line leaves the XCD's L2 as it is written instead of staying dirty until the
// end-of-kernel release writes the whole output back in one piece (MI355X_MICROARCH.md, "stores of each flavour").
// Through the buffer-store builtin (cache policy bit 4 = sc1 on gfx94x/95x), so that the compiler tracks the store in
// vmcnt: the first version issued it from inline assembly, which let the compiler overwrite the data registers while
// the store was still reading them -- right or wrong by scheduling luck (caught by tests/test_gpu_gemm.py on the
// variants build).
__device__ __forceinline__ void store16_wt(float* base, unsigned elem_off, const float4& v) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffff, 0x00020000);
    sert_f4 x = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)(elem_off * 4u), 0, /*sc1*/ 1 << 4);
}

struct StreamGemmArgs {
    const float* A;      // (M, K) row-major, lda
    const float* B;      // BRC: (K, N) row-major, ldb   else: (N, K) row-major, ldb
    float* C;            // (M, N) row-major, ldc
    const float* bias;   // (N) or null
    int M, lda, ldb, ldc;
    int nstrips;         // cdiv(M, 16)
    int ko;              // (experiments) 1: no MFMAs, 2: no stores, 4: no A loads after the first strip
};

#ifndef SERT_STREAM_WAVES
#define SERT_STREAM_WAVES 8
#endif
constexpr int kStreamWaves = SERT_STREAM_WAVES;

template <bool BRC, int EPI, int NB, int KB>
__global__ __launch_bounds__(64 * kStreamWaves, 8 / kStreamWaves) void gemm_stream_f32(const StreamGemmArgs g) {
    constexpr int N = 16 * NB, K = 16 * KB;
    constexpr int LDB = BRC ? N : K + 4;              // (N, K) rows padded: 16 consecutive rows hit 16 distinct bank quads
    constexpr int BROWS = BRC ? K : N;
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [BROWS][LDB]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
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
    if (s < g.nstrips) load_strip(s, a);          // (in flight while B is copied)
    // ---- B -> LDS, memory layout = LDS layout: 16-byte pieces, no transposition ----
    {
        constexpr int PIECES_PER_ROW = (BRC ? N : K) / 4;
        for (int p = threadIdx.x; p < BROWS * PIECES_PER_ROW; p += 64 * kStreamWaves) {
            const int r = p / PIECES_PER_ROW, cc = (p - r * PIECES_PER_ROW) * 4;
            *reinterpret_cast<float4*>(&Bs[r * LDB + cc]) = *reinterpret_cast<const float4*>(g.B + (size_t)r * g.ldb + cc);
        }
    }
    // bias of this lane's columns
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = 0.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = g.bias[BRC ? 64 * (nb >> 2) + 4 * c + (nb & 3) : 16 * nb + c];
    }
    __syncthreads();
    // The two waves of a SIMD (w and w + 4) would run in lockstep -- MFMAs at the same time (sharing the pipe), epilogues
    // at the same time (the pipe idle): the second one starts half a unit late, so that one's tanh / store epilogue
    // falls into the other's MFMA phase, and the shared pipe keeps them half a unit apart from then on.
    if ((g.ko & 32) == 0 && wv >= kStreamWaves / 2) __builtin_amdgcn_s_sleep(32);      // ~2000 cycles
    // A work unit is (strip, 64-column half): 4 K MFMAs into SIXTEEN accumulator registers, then their epilogue.  The two
    // halves of a strip use different accumulators, so the stores of one half are in flight under the MFMAs of the next
    // (a store holds its source registers until it has completed -- vmcnt counts stores on this part): with whole
    // strips per accumulator set, every wave of the chip stored at the same moments and the matrix pipe idled for them
    // (29.7 us; MFMA loop alone 21.4, loads + stores alone 13-15).
    while (s < g.nstrips) {
        const int sn = s + stride;
        float4 an[KB];
        if (sn < g.nstrips) { if (g.ko & 4) { for (int j = 0; j < KB; ++j) an[j] = a[j]; } else load_strip(sn, an); }
        const int row0 = s * 16 + 4 * q;
#pragma unroll
        for (int h = 0; h < NB / 4; ++h) {
            f32x4v acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = (f32x4v)(0.f);
            if (g.ko & 1) {
#pragma unroll
                for (int j = 0; j < KB; ++j) { acc[j % 4][0] += a[j].x; acc[j % 4][1] += a[j].y; acc[j % 4][2] += a[j].z; acc[j % 4][3] += a[j].w; }
            } else if (BRC) {
                // K / 4 steps of one 16-byte read (four column blocks) and four MFMAs; the read runs one step ahead
                float4 bcur = *reinterpret_cast<const float4*>(&Bs[(4 * q) * LDB + 64 * h + 4 * c]), bnxt = bcur;
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 4 * j + i;
                        if (t + 1 < 4 * KB) {
                            const int kn = 16 * ((t + 1) >> 2) + 4 * q + ((t + 1) & 3);
                            bnxt = *reinterpret_cast<const float4*>(&Bs[kn * LDB + 64 * h + 4 * c]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur.x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur.y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur.z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bcur.w, acc[3], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bcur = bnxt;
                    }
                }
            } else {
                // K / 16 blocks of four 16-byte reads (one per column block, four k-steps each) and sixteen MFMAs
                float4 bcur[4], bnxt[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) bcur[b] = *reinterpret_cast<const float4*>(&Bs[(16 * (4 * h + b) + c) * LDB + 4 * q]);
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
                    if (j + 1 < KB) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) bnxt[b] = *reinterpret_cast<const float4*>(&Bs[(16 * (4 * h + b) + c) * LDB + 16 * (j + 1) + 4 * q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bcur[b].x, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bcur[b].y, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bcur[b].z, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bcur[b].w, acc[b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) bcur[b] = bnxt[b];
                }
            }
            // ---- epilogue of the half: D[row = 4 q + r][column of (block b, lane c)] = acc[b][r] ----
            if ((g.ko & 2) && acc[0][0] != 123.456f) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = row0 + r;
                if (row >= g.M) continue;
                if (g.ko & 16) row &= 1023;       // (experiment: every store lands in the first 512 KB of C)
                if (BRC) {
                    float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
                        v.x += bv[4 * h + 0]; v.y += bv[4 * h + 1]; v.z += bv[4 * h + 2]; v.w += bv[4 * h + 3];
                    }
                    if (EPI == EPI_BIAS_TANH) { v.x = fast_tanh(v.x); v.y = fast_tanh(v.y); v.z = fast_tanh(v.z); v.w = fast_tanh(v.w); }
                    if (g.ko & 8) *reinterpret_cast<float4*>(g.C + (size_t)row * g.ldc + 64 * h + 4 * c) = v;
                    else store16_wt(g.C, (unsigned)row * (unsigned)g.ldc + 64u * h + 4u * c, v);     // (M * ldc < 2^30 here)
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float v = acc[b][r];
                        if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) v += bv[4 * h + b];
                        if (EPI == EPI_BIAS_TANH) v = fast_tanh(v);
                        g.C[(size_t)row * g.ldc + 16 * (4 * h + b) + c] = v;
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
    if (N != 128 || K != 128 || M < 8192 || (long long)M * ldc >= (1ll << 30)) return false;                      // (the instantiated shape: d_w = d_e = 128)
    if (lda % 4 || ldb % 4 || ldc % 4 || ((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16) return false;
    StreamGemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.nstrips = cdiv(M, 16);
    g.ko = variant_knob("SERT_STREAM_KO") ? atoi(variant_knob("SERT_STREAM_KO")) : 0;
    constexpr bool BRC = !TB;
    constexpr int E = (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) ? EPI : EPI_STORE;
    const size_t lds = (size_t)(BRC ? 128 * 128 : 128 * 132) * sizeof(float);
    auto kern = gemm_stream_f32<BRC, E, 8, 8>;
    static const bool attr_ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!attr_ok) return false;
    const int grid = std::min(256 * (8 / kStreamWaves), cdiv(g.nstrips, kStreamWaves));
    SERT_LAUNCH(kern, dim3(grid), dim3(64 * kStreamWaves), lds, s, g);
    return true;
}

}  // namespace sert
