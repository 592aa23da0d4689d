// Large-tile fp32 MFMA GEMM for gfx950: C (M,N) = A (M,K) . B^T, A and B row-major with K
// contiguous ("NT": scoring P.E^T, dG = dZ.W^T with W stored (d, V) ... ).
//
// Why a second kernel: gemm.h's 64x64 per-wave tile needs one LDS fragment read per MFMA
// and tops out at ~67 % of the fp32 MFMA peak however it is scheduled (knock-outs,
// occupancy sweep).  The vendor library reaches 95 % with ONE wave per SIMD owning a
// 128x128 tile (hipBLASLt MT256x256x32 MI16x16): 16-byte LDS reads each feeding four
// 16x16x4 MFMA steps -- 1/16 of a read per MFMA -- so the matrix pipe is never waiting on
// anything this wave does.  This kernel takes that shape:
//   workgroup 256 threads = 2x2 waves, tile 256x256, K slab 16
//   wave tile 128x128 = 8x8 blocks of v_mfma_f32_16x16x4_f32, 256 accumulator registers
//   LDS: operands stored [row][16 k] (row stride 20 floats), double buffered, 80 KB
//   a lane's fragment read = float4 = k-quad (lane/16) of row (lane%16) of a 16-row block;
//   MFMA step j of the slab contracts the k's {4g + j}: any partition of k is a valid order
#pragma once
#include "../common.h"

namespace sert {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 16, BLD = 20;

struct BigGemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n;
    // FILTER (query scoring; same contract as gemm.h's EPI_FILTER): nothing is stored to C;
    // the elements of row r that reach thr[r] go, as (order-preserving key, column), to the
    // list of their (row, 64-column group): cand[(r * ngr + group) * cap + slot], cnt = count
    const float* thr;
    unsigned long long* cand;
    unsigned char* cnt;       // zeroed by the caller
    int ngr, cap;
};

// Any M, N (edge tiles load clamped rows and drop them in the epilogue); K % 16 == 0 and
// 16-byte aligned rows (lda, ldb % 4 == 0).  The tile index runs along M first: the 256
// workgroups in flight share a handful of B tiles and all of A.
//
// NBJ = 16-column blocks per wave: 8 -> tile 256x256, wave 128x128, 256 accumulators, one wave per
// SIMD (gemm_big_nt); 4 -> tile 256x128, wave 128x64, 128 accumulators, two workgroups per CU
// (gemm_mid_nt): half the MFMAs per fragment byte, but a second workgroup's MFMAs run under
// this one's barrier, LDS stores and -- for FILTER -- its long epilogue.
template <bool FILTER, int NBJ>
__device__ __forceinline__ void gemm_big_body(const BigGemmArgs& g) {
    constexpr int BNT = 32 * NBJ;          // tile columns: 2 waves x NBJ x 16
    __shared__ __attribute__((aligned(16))) float As[2][BM][BLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BNT][BLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x;
    const int tn = tile / g.tiles_m, tm = tile - tn * g.tiles_m;
    const int m0 = tm * BM, n0 = tn * BNT;

    // global -> registers: 256 rows x 4 k-quads per operand = 1024 float4, 4 per thread:
    // thread t loads k-quad (t & 3) of rows (t >> 2) + 64 i
    // (straight-line macros, not lambdas: arrays captured by reference are not scalarised and
    //  hipcc's promote-alloca pass then parks them in LDS -- a load -> LDS -> LDS -> LDS prefetch)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const int lq = tid & 3, lrow = tid >> 2;
#define BIG_ROWPTR(X, ld, r0, R, i) ((X) + (size_t)min((r0) + lrow + 64 * (i), (R) - 1) * (ld) + 4 * lq)
    const float* Ap0 = BIG_ROWPTR(g.A, g.lda, m0, g.M, 0);
    const float* Ap1 = BIG_ROWPTR(g.A, g.lda, m0, g.M, 1);
    const float* Ap2 = BIG_ROWPTR(g.A, g.lda, m0, g.M, 2);
    const float* Ap3 = BIG_ROWPTR(g.A, g.lda, m0, g.M, 3);
    const float* Bp0 = BIG_ROWPTR(g.B, g.ldb, n0, g.N, 0);
    const float* Bp1 = BIG_ROWPTR(g.B, g.ldb, n0, g.N, 1);
    const float* Bp2 = BIG_ROWPTR(g.B, g.ldb, n0, g.N, 2);
    const float* Bp3 = BIG_ROWPTR(g.B, g.ldb, n0, g.N, 3);
#undef BIG_ROWPTR
#define BIG_GLOAD(k0)                                                         \
    do {                                                                      \
        ra0 = *reinterpret_cast<const float4*>(Ap0 + (k0));                   \
        ra1 = *reinterpret_cast<const float4*>(Ap1 + (k0));                   \
        ra2 = *reinterpret_cast<const float4*>(Ap2 + (k0));                   \
        ra3 = *reinterpret_cast<const float4*>(Ap3 + (k0));                   \
        rb0 = *reinterpret_cast<const float4*>(Bp0 + (k0));                   \
        rb1 = *reinterpret_cast<const float4*>(Bp1 + (k0));                   \
        if (NBJ == 8) {                                                       \
            rb2 = *reinterpret_cast<const float4*>(Bp2 + (k0));               \
            rb3 = *reinterpret_cast<const float4*>(Bp3 + (k0));               \
        }                                                                     \
    } while (0)
#define BIG_LSTORE(buf)                                                       \
    do {                                                                      \
        *reinterpret_cast<float4*>(&As[buf][lrow][4 * lq]) = ra0;             \
        *reinterpret_cast<float4*>(&As[buf][lrow + 64][4 * lq]) = ra1;        \
        *reinterpret_cast<float4*>(&As[buf][lrow + 128][4 * lq]) = ra2;       \
        *reinterpret_cast<float4*>(&As[buf][lrow + 192][4 * lq]) = ra3;       \
        *reinterpret_cast<float4*>(&Bs[buf][lrow][4 * lq]) = rb0;             \
        *reinterpret_cast<float4*>(&Bs[buf][lrow + 64][4 * lq]) = rb1;        \
        if (NBJ == 8) {                                                       \
            *reinterpret_cast<float4*>(&Bs[buf][(lrow + 128) % BNT][4 * lq]) = rb2; \
            *reinterpret_cast<float4*>(&Bs[buf][(lrow + 192) % BNT][4 * lq]) = rb3; \
        }                                                                     \
    } while (0)

    const int rbase = m0 + wr * 128 + 4 * lg, cbase = n0 + wc * (16 * NBJ) + lr;

    f32x4 acc[8][NBJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    BIG_GLOAD(0);
    BIG_LSTORE(0);
    __syncthreads();
    // Software pipeline over the two k-halves of a slab (k = 4 lg + {0,1} | {2,3}): the
    // fragment reads of one half are in flight under the 128 MFMAs of the other, and the
    // barrier + LDS store of the next slab sit between the halves, not at the loop edge.
    float2 fa0[8], fb0[NBJ], fa1[8], fb1[NBJ];
#define BIG_FRAG(fa, fb, buf, h)                                                                      \
    _Pragma("unroll") for (int b = 0; b < 8; ++b)                                                     \
        fa[b] = *reinterpret_cast<const float2*>(&As[buf][wr * 128 + b * 16 + lr][4 * lg + 2 * (h)]); \
    _Pragma("unroll") for (int b = 0; b < NBJ; ++b)                                                   \
        fb[b] = *reinterpret_cast<const float2*>(&Bs[buf][wc * (16 * NBJ) + b * 16 + lr][4 * lg + 2 * (h)]);
#define BIG_MFMA(fa, fb)                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
    _Pragma("unroll") for (int bi = 0; bi < 8; ++bi)                                                  \
    _Pragma("unroll") for (int bj = 0; bj < NBJ; ++bj)                                                \
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"                                          \
                     : "+a"(acc[bi][bj]) : "v"(j ? fa[bi].y : fa[bi].x), "v"(j ? fb[bj].y : fb[bj].x));
    int buf = 0;
    BIG_FRAG(fa0, fb0, 0, 0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        // (branch-free body: the last iteration re-fetches its own slab into the idle buffer;
        //  with the prefetch under `if (next)` the accumulators live across basic blocks and
        //  hipcc shuttles them between the VGPR and AGPR files, 2-3 moves per MFMA)
        const int kn = min(k0 + BK, g.K - BK);
        BIG_GLOAD(kn);
        BIG_FRAG(fa1, fb1, buf, 1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        BIG_MFMA(fa0, fb0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        BIG_LSTORE(buf ^ 1);
        __syncthreads();
        BIG_FRAG(fa0, fb0, buf ^ 1, 0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        BIG_MFMA(fa1, fb1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
    }
#undef BIG_FRAG
#undef BIG_MFMA
    // (asm MFMAs are opaque to the hazard recogniser: drain the pipe before reading acc)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + i
    if (!FILTER) {
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rbase + bi * 16 + i;
                if (row >= g.M) continue;
                float* Cr = g.C + (size_t)row * g.ldc;
#pragma unroll
                for (int bj = 0; bj < NBJ; ++bj)
                    if (cbase + bj * 16 < g.N) Cr[cbase + bj * 16] = acc[bi][bj][i];
            }
    } else {
        // The 16 lanes of a quarter-wave hold 16 consecutive columns of one row, so an element's
        // slot in its group list is a ballot/popcount prefix over the group's four 16-column
        // blocks: no atomics, deterministic (ascending column) order.
        float th[8][4];      // the lane's 32 row thresholds, one batch of loads
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rbase + bi * 16 + i;
                const float t = g.thr[min(row, g.M - 1)];
                th[bi][i] = row < g.M ? t : INFINITY;
            }
        if (n0 + BNT > g.N) {   // edge tile: the (clamped, duplicated) columns beyond N never pass
#pragma unroll
            for (int bj = 0; bj < NBJ; ++bj)
                if (cbase + bj * 16 >= g.N) {
#pragma unroll
                    for (int bi = 0; bi < 8; ++bi)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[bi][bj][i] = -INFINITY;
                }
        }
        // The no-candidate case is compare + branch; all addresses are one 64-bit lane base + a
        // 32-bit index built from uniform strides.
        const unsigned below = (1u << lr) - 1u, sh = 16u * lg;
        const unsigned ucap = (unsigned)g.cap, ngr = (unsigned)g.ngr, rstride = ngr * ucap;
        const unsigned g0 = (unsigned)(n0 / 64 + wc * (NBJ / 4));
        unsigned long long* cand_lane = g.cand + ((size_t)rbase * ngr + g0) * ucap;
        unsigned char* cnt_lane = g.cnt + (size_t)rbase * ngr + g0;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ro = (unsigned)(bi * 16 + i);
                const float t = th[bi][i];
                unsigned run = 0;
#pragma unroll
                for (int bj = 0; bj < NBJ; ++bj) {
                    if ((bj & 3) == 0) run = 0;
                    const float v = acc[bi][bj][i];
                    const bool p = v >= t;
                    const unsigned long long any = __builtin_amdgcn_ballot_w64(p);
                    if (any) {     // ~2/3 of the (4 row x 16 column) blocks hold no candidate at all
                        const unsigned h = (unsigned)(any >> sh) & 0xffffu;
                        if (p) {
                            const unsigned slot = run + __popc(h & below);
                            if (slot < ucap)
                                cand_lane[ro * rstride + (unsigned)(bj >> 2) * ucap + slot] =
                                    ((unsigned long long)desc_key(v) << 32) | (unsigned)(cbase + bj * 16);
                        }
                        run += __popc(h);
                    }
                    if ((bj & 3) == 3 && run && lr == 0)
                        cnt_lane[ro * ngr + (unsigned)(bj >> 2)] = (unsigned char)(run > 255u ? 255u : run);
                }
            }
    }
}

#undef BIG_GLOAD
#undef BIG_LSTORE

template <bool FILTER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_big_nt(const BigGemmArgs g) {
    gemm_big_body<FILTER, 8>(g);
}
template <bool FILTER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_mid_nt(const BigGemmArgs g) {
    gemm_big_body<FILTER, 4>(g);
}

inline bool gemm_big_ok(int K, int lda, int ldb) { return K >= BK && K % BK == 0 && lda % 4 == 0 && ldb % 4 == 0; }

// mid = true: 256x128 tiles, two workgroups per CU
inline void launch_gemm_big_nt(hipStream_t s, const float* A, const float* B, float* C, int M, int N, int K,
                               int lda, int ldb, int ldc, bool mid = false) {
    BigGemmArgs g = {};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(N, mid ? 128 : 256);
    if (mid) hipLaunchKernelGGL(gemm_mid_nt<false>, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_big_nt<false>, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
}

// rows of A that reach thr[row] -> (row, 64-column group) candidate lists; ngr groups per row
inline void launch_gemm_big_filter(hipStream_t s, const float* A, const float* B, const float* thr,
                                   unsigned long long* cand, unsigned char* cnt, int ngr, int cap,
                                   int M, int N, int K, int lda, int ldb, bool mid = false) {
    BigGemmArgs g = {};
    g.A = A; g.B = B; g.C = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = 0;
    g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(N, mid ? 128 : 256);
    g.thr = thr; g.cand = cand; g.cnt = cnt; g.ngr = ngr; g.cap = cap;
    if (mid) hipLaunchKernelGGL(gemm_mid_nt<true>, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_big_nt<true>, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
}

}  // namespace sert
