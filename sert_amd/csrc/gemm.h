// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, == an fmaf chain).
//
// Used for the only dense contractions on the path:
//   vectorspace  a = h.W + b (tanh epilogue)      sert/models.py:1057-1061
//                dW = h^T.da ; dh = da.W^T        (autodiff of the above)
//   loglinear    Z = G.W + b                      sert/models.py:846-849
//                dW = G^T.dZ ; dG = dZ.W^T
//   scoring      S = Q.E^T                        bin/query.py:352-357 (batched)
//
// Workgroup = 256 threads = 4 waves (2x2); tile 128x128x32; each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  Operands are
// staged through LDS k-major ([k][m] / [k][n], leading dim 132) so that a
// wave's MFMA fragment read (32 consecutive m or n for one k) is one
// conflict-free ds_read_b32 per 32-lane half.
//
// The shapes on this path are skinny (K = 128 for the projections, or one
// 128x128 output with K = batch for dW), so what decides the rate is not LDS
// bandwidth (a 32x32x2 fp32 MFMA occupies its SIMD for 64 cycles) but pipeline
// bubbles.  Hence: (1) register-staged double buffering -- the global loads of
// k-slab t+1 are issued before the MFMAs of slab t and written to the other LDS
// buffer after them, one barrier per slab; (2) persistent workgroups that walk
// a flat (tile, k-slab) sequence, so the prefetch crosses tile boundaries and
// the epilogue stores of a tile overlap the loads of the next; (3) two
// workgroups per CU (67.6 KB LDS each) so one computes while the other waits.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace sert {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SERT_GK
#define SERT_GK 16
#endif
#ifndef SERT_GEMM_WAVES
#define SERT_GEMM_WAVES 4
#endif
constexpr int GM = 128, GN = 128, GK = SERT_GK, GLD = 132;
constexpr int GNV = GK / 8;          // float4 pieces per thread per operand slab
constexpr int GTPR = 256 / GK;       // threads per k-row in the k-major loader

enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_TANH = 2, EPI_FILTER = 3, EPI_ACCUM = 4 };
// EPI_ACCUM: C += op(A).op(B) (read-modify-write of C; the caller orders the launches that add to one C)
// EPI_FILTER (query scoring): nothing is stored to C.  The elements of row r that reach
// the row's threshold (bias[r]) are written, as (order-preserving key, column), to small
// per-(row, 64-column group) lists: the 32 lanes of a half-wave hold the same row, so the
// slot of an element is a ballot/popcount prefix -- NO atomics (device-scope atomics run
// at ~4 G/s on this part: 8 M appends cost more than the whole GEMM).  cand holds
// `cap` slots per group, cnt (bytes) the group's element count (saturated at 255; a
// count > cap tells the consumer the group overflowed).  Deterministic layout.

// tanh for the projection epilogue (sert/models.py:1055): ~15 instructions instead
// of the libm expansion (which, inlined 64x per lane, spilled the accumulators).
//   |x| <  0.25 : odd Taylor series through x^9   (truncation < 3e-9 relative)
//   |x| >= 0.25 : 1 - 2/(exp(2|x|) + 1)           (<= ~4 ulp; saturates to 1 for |x| > 9)
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x);
    const float x2 = x * x;
    const float poly = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.05396825f + x2 * 0.02186949f))));
    // exp(2|x|) = exp2(2 log2(e) |x|); 2 / (e + 1) through the hardware reciprocal (1 ulp) instead of
    // the ~10-instruction IEEE division: the epilogue's VALU work is not hidden behind the MFMAs of
    // these short-K GEMMs (gemm_strip.h), 64 tanh per lane and tile: 34.5 -> 32.0 us at C2
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * fminf(ax, 10.0f));
    const float big = copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f), x);
    return ax < 0.25f ? poly : big;
}

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int kper;                 // k range per split (multiple of GK)
    int splits;
    int tiles_m, tiles_n;
    size_t c_split_stride;    // C offset between splits
    int vecA, vecB;           // 16-byte loads allowed
    // EPI_FILTER only
    unsigned long long* cand;   // [M][2*tiles_n][cap]
    unsigned char* cnt;         // [M][2*tiles_n], zeroed by the caller
    int cap;
    // 64x64-tile kernel only (optional): output row r is stored as row rowmap[r] of C -- the loglinear dG, one row
    // per distinct word, lands straight on the word-table gradient rows (no scatter pass)
    const int32_t* rowmap = nullptr;
};

// --- global -> registers (zero padded) ----------------------------------------
// VEC = true : every 4-float piece is 16-byte aligned and either fully inside or
//              fully outside the matrix (leading dims, extents and pointers are
//              multiples of 4 floats) -> branch-free: one float4 load from a
//              clamped (always valid) address; bit j of `mask` says whether piece j
//              is inside.  The zeroing happens in lstore_*, AFTER the MFMAs of the
//              current slab: touching the loaded value here (even a multiply by
//              the mask) makes the compiler wait for the load on the spot, which
//              turns the prefetch into a synchronous load (73 % -> MFMA busy).
// VEC = false: scalar guarded loads (odd sizes; small problems only).
//
// One 16-byte piece of a tile: buffer_load_dwordx4 -- a 128-bit resource descriptor in SGPRs (the workgroup-uniform
// tile base) + a 32-bit lane offset -- instead of global_load_dwordx4 on a 64-bit address pair per piece.  Same
// bytes; the 128 x 160 kernel (168 VGPRs) gains 7-8 % from the shorter address path (C4 projection 158 -> 147 us,
// dh 155 -> 143 us), the 128 x 128 one measures the same.  Offsets stay far below 2^31 bytes: launch_gemm takes the
// 16-byte path only for leading dimensions below 2^22 elements.  SERT_GEMM_GLOBAL_LOADS restores the plain loads.
typedef float sert_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 tile_load16(const float* base, unsigned elem_off) {
#ifndef SERT_GEMM_GLOBAL_LOADS
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xffffffff, 0x00020000);
    const sert_f4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem_off * 4u), 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(base + elem_off);
#endif
}
// Source stored [k][c], contiguous along c (the tile's M or N axis).
// `base` = src + k0*ld + c0 is workgroup-uniform; lane offsets are 32-bit.
template <bool VEC>
__device__ __forceinline__ void gload_kmajor(const float* __restrict__ base, int ld, int krem,
                                             int crem, float4 (&r)[GNV], unsigned& mask) {
    mask = 0xffffffffu;
    const int t = threadIdx.x;
    const int kr = t / GTPR;
    const int cq = (t % GTPR) * 4;
    const bool kok = kr < krem;
    const unsigned roff = kok ? (unsigned)kr * (unsigned)ld : 0u;
#pragma unroll
    for (int j = 0; j < GNV; ++j) {
        const int c = cq + j * (GTPR * 4);
        if (VEC) {
            const bool ok = kok && (c < crem);
            if (!ok) mask &= ~(1u << j);
            r[j] = tile_load16(base, ok ? roff + (unsigned)c : 0u);
        } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok) {
                if (c + 0 < crem) v.x = base[roff + c + 0];
                if (c + 1 < crem) v.y = base[roff + c + 1];
                if (c + 2 < crem) v.z = base[roff + c + 2];
                if (c + 3 < crem) v.w = base[roff + c + 3];
            }
            r[j] = v;
        }
    }
}
__device__ __forceinline__ void lstore_kmajor(float (*dst)[GLD], const float4 (&r)[GNV], unsigned mask) {
    const int t = threadIdx.x;
    const int kr = t / GTPR, cq = (t % GTPR) * 4;
#pragma unroll
    for (int j = 0; j < GNV; ++j) {
        const bool ok = (mask >> j) & 1u;
        *reinterpret_cast<float4*>(&dst[kr][cq + j * (GTPR * 4)]) =
            make_float4(ok ? r[j].x : 0.f, ok ? r[j].y : 0.f, ok ? r[j].z : 0.f, ok ? r[j].w : 0.f);
    }
}

// Source stored [c][k], contiguous along k: transposed on the way into LDS.
// `base` = src + c0*ld + k0 is workgroup-uniform.
template <bool VEC>
__device__ __forceinline__ void gload_cmajor(const float* __restrict__ base, int ld, int krem,
                                             int crem, float4 (&r)[GNV], unsigned& mask) {
    mask = 0xffffffffu;
    const int t = threadIdx.x;
    const int c = t >> 1;
    const int kh = (t & 1) * (GK / 2);
    const bool cok = c < crem;
    const unsigned roff = cok ? (unsigned)c * (unsigned)ld : 0u;
#pragma unroll
    for (int j = 0; j < GNV; ++j) {
        const int k = kh + j * 4;
        if (VEC) {
            const bool ok = cok && (k < krem);
            if (!ok) mask &= ~(1u << j);
            r[j] = tile_load16(base, ok ? roff + (unsigned)k : 0u);
        } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok) {
                if (k + 0 < krem) v.x = base[roff + k + 0];
                if (k + 1 < krem) v.y = base[roff + k + 1];
                if (k + 2 < krem) v.z = base[roff + k + 2];
                if (k + 3 < krem) v.w = base[roff + k + 3];
            }
            r[j] = v;
        }
    }
}
__device__ __forceinline__ void lstore_cmajor(float (*dst)[GLD], const float4 (&r)[GNV], unsigned mask) {
    const int t = threadIdx.x;
    const int c = t >> 1, kh = (t & 1) * (GK / 2);
#pragma unroll
    for (int j = 0; j < GNV; ++j) {
        const int kl = kh + j * 4;
        const bool ok = (mask >> j) & 1u;
        dst[kl + 0][c] = ok ? r[j].x : 0.f;
        dst[kl + 1][c] = ok ? r[j].y : 0.f;
        dst[kl + 2][c] = ok ? r[j].z : 0.f;
        dst[kl + 3][c] = ok ? r[j].w : 0.f;
    }
}

// C[z] (M,N) = epi( op(A) (M,K) . op(B) (K,N) ) over the K range of split z.
//   TA=false: A row-major (M,K)      TA=true: A stored (K,M)  -> computes A^T.B
//   TB=false: B row-major (K,N)      TB=true: B stored (N,K)  -> computes A.B^T
// Work item w in [0, tiles_m*tiles_n*splits): split z = w / (tiles_m*tiles_n),
// tile = w % (..), tile row = tile / tiles_n.  Split z covers k in
// [z*kper, min(K,(z+1)*kper)) and writes to C + z*c_split_stride.
// CSB: additionally emit the column sums of op(B) over this split's k range
// (row tile 0 only) at C[z] + M*N .. +N  -- used for db = sum_i da_i, which
// rides along with the dW = h^T.da GEMM for free (the da tile is in LDS anyway).
// FULL: M and N are multiples of the tile => no guards in the epilogue.
template <bool TA, bool TB, int EPI, bool CSB, bool VEC, bool FULL>
__global__ __launch_bounds__(256, SERT_GEMM_WAVES) void gemm_f32_mfma(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GLD];
    __shared__ float cs_lds[GN];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_mn = g.tiles_m * g.tiles_n;
    const int total = tiles_mn * g.splits;

    int item = blockIdx.x;
    if (item >= total) return;

    int m0, n0, kbeg, kend, z, tm_idx;
    auto decode = [&](int it) {
        z = it / tiles_mn;
        const int t = it - z * tiles_mn;
        tm_idx = t / g.tiles_n;
        m0 = tm_idx * GM;
        n0 = (t - tm_idx * g.tiles_n) * GN;
        kbeg = z * g.kper;
        kend = min(g.K, kbeg + g.kper);
    };
    unsigned mka = 0xffffffffu, mkb = 0xffffffffu;   // inside-masks of the staged pieces
    auto gload = [&](int mm0, int nn0, int k0, int ke, float4 (&ra)[GNV], float4 (&rb)[GNV]) {
        // tile base pointers are workgroup-uniform (SGPRs); per-lane offsets 32-bit.
        // The leading dimensions are made opaque here so that the lane offsets are
        // recomputed per slab (a handful of VALU ops) instead of being hoisted out
        // of the persistent loop and spilled (64-bit addresses x 16 loads).
        int lda_ = g.lda, ldb_ = g.ldb;
        asm volatile("" : "+s"(lda_), "+s"(ldb_));
        if (TA) gload_kmajor<VEC>(g.A + (size_t)k0 * lda_ + mm0, lda_, ke - k0, g.M - mm0, ra, mka);
        else    gload_cmajor<VEC>(g.A + (size_t)mm0 * lda_ + k0, lda_, ke - k0, g.M - mm0, ra, mka);
        if (TB) gload_cmajor<VEC>(g.B + (size_t)nn0 * ldb_ + k0, ldb_, ke - k0, g.N - nn0, rb, mkb);
        else    gload_kmajor<VEC>(g.B + (size_t)k0 * ldb_ + nn0, ldb_, ke - k0, g.N - nn0, rb, mkb);
    };
    auto lstore = [&](int buf, const float4 (&ra)[GNV], const float4 (&rb)[GNV]) {
        if (TA) lstore_kmajor(As[buf], ra, mka); else lstore_cmajor(As[buf], ra, mka);
        if (TB) lstore_cmajor(Bs[buf], rb, mkb); else lstore_kmajor(Bs[buf], rb, mkb);
    };

    // bias of this wave's two column groups, fetched at the START of every tile so
    // that its latency hides under the MFMAs (loaded in the epilogue it cost ~6 us)
    float bias_v[2] = {0.f, 0.f};
    auto load_bias = [&]() {
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = n0 + wc * 64 + tn * 32 + li;
                bias_v[tn] = (col < g.N) ? g.bias[col] : 0.f;
            }
        }
    };
    float4 ra[GNV], rb[GNV];
    decode(item);
    load_bias();
    gload(m0, n0, kbeg, kend, ra, rb);
    lstore(0, ra, rb);
    __syncthreads();
    int buf = 0;

    f32x16 acc[2][2];
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    while (true) {
        const int next_item = item + gridDim.x;
        const bool has_next_item = next_item < total;
        for (int k0 = kbeg; k0 < kend; k0 += GK) {
            const bool next_k = (k0 + GK) < kend;
            bool staged = false;
            // ---- prefetch the next k-slab (of this tile, or slab 0 of the next) ----
            if (next_k) {
                gload(m0, n0, k0 + GK, kend, ra, rb);
                staged = true;
            } else if (has_next_item) {
                const int zz = next_item / tiles_mn;
                const int t = next_item - zz * tiles_mn;
                const int tmi = t / g.tiles_n;
                const int kb = zz * g.kper;
                gload(tmi * GM, (t - tmi * g.tiles_n) * GN, kb, min(g.K, kb + g.kper), ra, rb);
                staged = true;
            }
            // ---- MFMAs on the current slab ----
            if (CSB && tm_idx == 0) {
                const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
                float cs = 0.f;
#pragma unroll
                for (int kk = 0; kk < GK / 2; ++kk) cs += Bs[buf][half * (GK / 2) + kk][col];
                csum += cs;
            }
            // Fragment reads one k-step AHEAD of the MFMAs that use them: left to itself the compiler puts each
            // step's ds_reads straight in front of its four MFMAs behind an s_waitcnt lgkmcnt(0), and the wave
            // sits out the LDS latency eight times per slab (round 3, from the ISA).  Worth little where two to
            // four waves per SIMD cover for each other (4096^3: 111.9 -> 113.6 TF), 8 % on the long-K, two-tiles-
            // per-CU shape of the full-softmax dp GEMM (65536 x 128 x 1000: 194 -> 179 us).
            float fa0 = As[buf][lh][wr * 64 + li], fa1 = As[buf][lh][wr * 64 + 32 + li];
            float fb0 = Bs[buf][lh][wc * 64 + li], fb1 = Bs[buf][lh][wc * 64 + 32 + li];
#pragma unroll
            for (int kk = 0; kk < GK; kk += 2) {
                const float a0 = fa0, a1 = fa1, b0 = fb0, b1 = fb1;
                if (kk + 2 < GK) {
                    const int k = kk + 2 + lh;
                    fa0 = As[buf][k][wr * 64 + li];
                    fa1 = As[buf][k][wr * 64 + 32 + li];
                    fb0 = Bs[buf][k][wc * 64 + li];
                    fb1 = Bs[buf][k][wc * 64 + 32 + li];
                }
                __builtin_amdgcn_sched_barrier(0);   // (keep the reads above the MFMAs below)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            // ---- stage the prefetched slab into the other buffer ----
            if (staged) lstore(buf ^ 1, ra, rb);

            if (!next_k) {
                // ---- epilogue of this tile (its stores overlap the next tile's loads) ----
                // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                float* Cz = g.C + (size_t)z * g.c_split_stride;
                if (CSB && tm_idx == 0) {
                    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
                    if (half == 1) cs_lds[col] = csum;
                    __syncthreads();
                    if (half == 0 && n0 + col < g.N)
                        Cz[(size_t)g.M * g.N + n0 + col] = csum + cs_lds[col];
                    csum = 0.f;
                }
                // tile base is workgroup-uniform; lane offsets inside the tile are 32-bit
                float* Ct = Cz + (size_t)m0 * g.ldc + n0;
                const int mrem = g.M - m0, nrem = g.N - n0;
                unsigned uld = (unsigned)g.ldc;
                asm volatile("" : "+s"(uld));   // keep the 64 store offsets out of the main loop's live set
                // pin the prefetched bias in registers NOW: otherwise every predicated
                // store block gets its own s_waitcnt vmcnt(0), which (vmcnt counts
                // stores on CDNA4) serialises the 64 stores behind each other
                float bv0 = bias_v[0], bv1 = bias_v[1];
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) asm volatile("" : "+v"(bv0), "+v"(bv1));
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const int col = wc * 64 + tn * 32 + li;
                        const int row0 = wr * 64 + tm * 32 + 4 * lh;
                        const float bv = tn ? bv1 : bv0;
                        unsigned off = (unsigned)row0 * uld + (unsigned)col;
                        if (EPI == EPI_FILTER) {
                            // handled once per tm (needs both tn values of the lane).  All
                            // addresses are a workgroup-uniform tile base + a 32-bit lane offset
                            // built from an opaque stride, so nothing is hoisted out of the
                            // persistent loop and spilled (cf. the C-store path).
                            if (tn == 0) {
                                const int col0 = col, col1 = col + 32;
                                const bool c0ok = col0 < nrem, c1ok = col1 < nrem;
                                const unsigned below = (1u << li) - 1u;
                                unsigned ngr = 2u * (unsigned)g.tiles_n;          // groups per row
                                unsigned ucap = (unsigned)g.cap;
                                asm volatile("" : "+s"(ngr), "+s"(ucap));
                                const size_t gbase = (size_t)m0 * ngr + 2u * (unsigned)(n0 / GN);
                                unsigned long long* cand_t = g.cand + gbase * ucap;
                                unsigned char* cnt_t = g.cnt + gbase;
                                const float* thr_t = g.bias + m0;
                                float thr[16];
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int lrow = row0 + (r & 3) + 8 * (r >> 2);
                                    thr[r] = (lrow < mrem) ? thr_t[lrow] : INFINITY;
                                }
                                unsigned goff = (unsigned)row0 * ngr + (unsigned)wc;   // group of (row, wc)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float v0 = acc[tm][0][r], v1 = acc[tm][1][r];
                                    const bool p0 = c0ok && v0 >= thr[r];
                                    const bool p1 = c1ok && v1 >= thr[r];
                                    const unsigned h0 = (unsigned)(__ballot(p0) >> (32 * lh));
                                    const unsigned h1 = (unsigned)(__ballot(p1) >> (32 * lh));
                                    if (h0 | h1) {
                                        const unsigned n0c = __popc(h0);
                                        if (p0) {
                                            const unsigned slot = __popc(h0 & below);
                                            if (slot < ucap)
                                                cand_t[goff * ucap + slot] =
                                                    ((unsigned long long)desc_key(v0) << 32) | (unsigned)(n0 + col0);
                                        }
                                        if (p1) {
                                            const unsigned slot = n0c + __popc(h1 & below);
                                            if (slot < ucap)
                                                cand_t[goff * ucap + slot] =
                                                    ((unsigned long long)desc_key(v1) << 32) | (unsigned)(n0 + col1);
                                        }
                                        if (li == 0) {
                                            const unsigned tot = n0c + __popc(h1);
                                            cnt_t[goff] = (unsigned char)(tot > 255u ? 255u : tot);
                                        }
                                    }
                                    goff += ((r & 3) == 3) ? 5u * ngr : ngr;
                                }
#pragma unroll
                                for (int r = 0; r < 16; ++r) { acc[tm][0][r] = 0.f; acc[tm][1][r] = 0.f; }
                            }
                        } else if (FULL) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float v = acc[tm][tn][r];
                                if (EPI == EPI_BIAS) v = v + bv;
                                if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bv);
                                if (EPI == EPI_ACCUM) v = v + Ct[off];
                                Ct[off] = v;
                                acc[tm][tn][r] = 0.f;
                                off += ((r & 3) == 3) ? 5u * uld : uld;
                            }
                        } else {
                            const bool cok = col < nrem;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = row0 + (r & 3) + 8 * (r >> 2);
                                float v = acc[tm][tn][r];
                                if (EPI == EPI_BIAS) v = v + bv;
                                if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bv);
                                if (cok && row < mrem) Ct[off] = (EPI == EPI_ACCUM) ? v + Ct[off] : v;
                                acc[tm][tn][r] = 0.f;
                                off += ((r & 3) == 3) ? 5u * uld : uld;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        if (!has_next_item) break;
        item = next_item;
        decode(item);
        load_bias();
    }
}

// ---- 128 x 160 tiles: N just above a multiple of 128 ------------------------------------------------
// d = 300 (C4, the reference's default --word_representation_size): three 128-column tiles cover 384
// columns for 300 -- 22 % of the MFMA work of the projection GEMMs and 39 % of dW's is padding.  Two
// 160-column tiles cover 320.  Four waves stacked along M, each a 32 x 160 strip = five 32x32x2
// accumulators (80 VGPRs); A side as above (128 rows), B side 160 columns (leading dimension 164).
// Same k order per output element as gemm_f32_mfma (one fmaf chain over k): bit-identical results.
// EPI_STORE / EPI_BIAS / EPI_BIAS_TANH, split-K and the column sums of op(B) (CSB); 16-byte loads only.
constexpr int GN2 = 160, GLD2 = 164;

__device__ __forceinline__ void gload_kmajor160(const float* __restrict__ base, int ld, int krem, int crem,
                                                float4 (&r)[3], unsigned& mask) {
    mask = 0x7u;
    const int t = threadIdx.x;
    const int kr = t / GTPR, cq = (t % GTPR) * 4;
    const bool kok = kr < krem;
    const unsigned roff = kok ? (unsigned)kr * (unsigned)ld : 0u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = cq + j * (GTPR * 4);
        const bool ok = kok && c < GN2 && c < crem;
        if (!ok) mask &= ~(1u << j);
        r[j] = tile_load16(base, ok ? roff + (unsigned)c : 0u);
    }
}
__device__ __forceinline__ void lstore_kmajor160(float (*dst)[GLD2], const float4 (&r)[3], unsigned mask) {
    const int t = threadIdx.x;
    const int kr = t / GTPR, cq = (t % GTPR) * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = cq + j * (GTPR * 4);
        if (c >= GN2) continue;
        const bool ok = (mask >> j) & 1u;
        *reinterpret_cast<float4*>(&dst[kr][c]) =
            make_float4(ok ? r[j].x : 0.f, ok ? r[j].y : 0.f, ok ? r[j].z : 0.f, ok ? r[j].w : 0.f);
    }
}
// source stored [c][k]: rows c = t / 2 (0..127) and, for the first 64 threads, 128 + t / 2
__device__ __forceinline__ void gload_cmajor160(const float* __restrict__ base, int ld, int krem, int crem,
                                                float4 (&r)[4], unsigned& mask) {
    mask = 0xfu;
    const int t = threadIdx.x;
    const int kh = (t & 1) * (GK / 2);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = p * 128 + (t >> 1);
        const bool cok = c < GN2 && c < crem && (p == 0 || t < 64);
        const unsigned roff = cok ? (unsigned)c * (unsigned)ld : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = kh + j * 4;
            const bool ok = cok && k < krem;
            if (!ok) mask &= ~(1u << (2 * p + j));
            r[2 * p + j] = tile_load16(base, ok ? roff + (unsigned)k : 0u);
        }
    }
}
__device__ __forceinline__ void lstore_cmajor160(float (*dst)[GLD2], const float4 (&r)[4], unsigned mask) {
    const int t = threadIdx.x;
    const int kh = (t & 1) * (GK / 2);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p == 1 && t >= 64) continue;
        const int c = p * 128 + (t >> 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kl = kh + j * 4;
            const bool ok = (mask >> (2 * p + j)) & 1u;
            const float4 v = r[2 * p + j];
            dst[kl + 0][c] = ok ? v.x : 0.f;
            dst[kl + 1][c] = ok ? v.y : 0.f;
            dst[kl + 2][c] = ok ? v.z : 0.f;
            dst[kl + 3][c] = ok ? v.w : 0.f;
        }
    }
}

template <bool TA, bool TB, int EPI, bool CSB>
__global__ __launch_bounds__(256, 3) void gemm_f32_mfma_n160(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GLD2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tiles_mn = g.tiles_m * g.tiles_n;
    const int total = tiles_mn * g.splits;
    int item = blockIdx.x;
    if (item >= total) return;
    int m0, n0, kbeg, kend, z, tm_idx;
    auto decode = [&](int it) {
        z = it / tiles_mn;
        const int t = it - z * tiles_mn;
        tm_idx = t / g.tiles_n;
        m0 = tm_idx * GM;
        n0 = (t - tm_idx * g.tiles_n) * GN2;
        kbeg = z * g.kper;
        kend = min(g.K, kbeg + g.kper);
    };
    unsigned mka = 0xffffffffu, mkb = 0xffffffffu;
    float4 ra[GNV];
    float4 rbk[3];
    float4 rbc[4];
    auto gload = [&](int mm0, int nn0, int k0, int ke) {
        int lda_ = g.lda, ldb_ = g.ldb;
        asm volatile("" : "+s"(lda_), "+s"(ldb_));
        if (TA) gload_kmajor<true>(g.A + (size_t)k0 * lda_ + mm0, lda_, ke - k0, g.M - mm0, ra, mka);
        else    gload_cmajor<true>(g.A + (size_t)mm0 * lda_ + k0, lda_, ke - k0, g.M - mm0, ra, mka);
        if (TB) gload_cmajor160(g.B + (size_t)nn0 * ldb_ + k0, ldb_, ke - k0, g.N - nn0, rbc, mkb);
        else    gload_kmajor160(g.B + (size_t)k0 * ldb_ + nn0, ldb_, ke - k0, g.N - nn0, rbk, mkb);
    };
    auto lstore = [&](int buf) {
        if (TA) lstore_kmajor(As[buf], ra, mka); else lstore_cmajor(As[buf], ra, mka);
        if (TB) lstore_cmajor160(Bs[buf], rbc, mkb); else lstore_kmajor160(Bs[buf], rbk, mkb);
    };
    float bias_v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_bias = [&]() {
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int col = n0 + j * 32 + li;
                bias_v[j] = (col < g.N) ? g.bias[col] : 0.f;
            }
        }
    };
    decode(item);
    load_bias();
    gload(m0, n0, kbeg, kend);
    lstore(0);
    __syncthreads();
    int buf = 0;
    f32x16 acc[5];
    float csum = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    while (true) {
        const int next_item = item + gridDim.x;
        const bool has_next_item = next_item < total;
        for (int k0 = kbeg; k0 < kend; k0 += GK) {
            const bool next_k = (k0 + GK) < kend;
            bool staged = false;
            if (next_k) {
                gload(m0, n0, k0 + GK, kend);
                staged = true;
            } else if (has_next_item) {
                const int zz = next_item / tiles_mn;
                const int t = next_item - zz * tiles_mn;
                const int tmi = t / g.tiles_n;
                const int kb = zz * g.kper;
                gload(tmi * GM, (t - tmi * g.tiles_n) * GN2, kb, min(g.K, kb + g.kper));
                staged = true;
            }
            if (CSB && tm_idx == 0 && threadIdx.x < GN2) {
                float cs = 0.f;
#pragma unroll
                for (int kk = 0; kk < GK; ++kk) cs += Bs[buf][kk][threadIdx.x];
                csum += cs;
            }
#pragma unroll
            for (int kk = 0; kk < GK; kk += 2) {
                const int k = kk + lh;
                const float a = As[buf][k][w * 32 + li];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float b = Bs[buf][k][j * 32 + li];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                }
            }
            if (staged) lstore(buf ^ 1);
            if (!next_k) {
                float* Cz = g.C + (size_t)z * g.c_split_stride;
                if (CSB && tm_idx == 0 && threadIdx.x < GN2) {
                    if (n0 + (int)threadIdx.x < g.N) Cz[(size_t)g.M * g.N + n0 + threadIdx.x] = csum;
                    csum = 0.f;
                }
                float* Ct = Cz + (size_t)m0 * g.ldc + n0;
                const int mrem = g.M - m0, nrem = g.N - n0;
                unsigned uld = (unsigned)g.ldc;
                asm volatile("" : "+s"(uld));
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int col = j * 32 + li;
                    const int row0 = w * 32 + 4 * lh;
                    float bv = bias_v[j];
                    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) asm volatile("" : "+v"(bv));
                    unsigned off = (unsigned)row0 * uld + (unsigned)col;
                    const bool cok = col < nrem;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + (r & 3) + 8 * (r >> 2);
                        float v = acc[j][r];
                        if (EPI == EPI_BIAS) v = v + bv;
                        if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bv);
                        if (cok && row < mrem) Ct[off] = (EPI == EPI_ACCUM) ? v + Ct[off] : v;
                        acc[j][r] = 0.f;
                        off += ((r & 3) == 3) ? 5u * uld : uld;
                    }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        if (!has_next_item) break;
        item = next_item;
        decode(item);
        load_bias();
    }
}

// ---- 64x64-tile variant for SMALL problems -------------------------------------------
// A GEMM with fewer 128x128 tiles than CUs (the projection and its backward at batch 4096:
// 32 x 3 tiles at d = 300) leaves most of the chip idle and every tile latency-bound: 35-40 us
// for 0.7 GFLOP.  Same arithmetic on 64x64 tiles (four waves of one 32x32 MFMA block each, the
// same k order, so the results are bit-identical to the kernel above): 4x the workgroups, one
// per tile, A row-major only, no split-K / column sums / filter.
constexpr int SM = 64, SLD = 68;

template <bool TB, int EPI, bool VEC>
__global__ __launch_bounds__(256, 4) void gemm_f32_mfma_small(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][SLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][SLD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
    const int m0 = tm * SM, n0 = tn * SM;
    // row-major operand (rows = tile rows, contiguous along k): thread -> row t / 4, k-quad t % 4
    const int cr = t >> 2, ck = (t & 3) * 4;
    // k-major operand (B stored (K, N)): thread -> k row t / 16, column quad t % 16
    const int kr = t >> 4, kc = (t & 15) * 4;
    auto load_rows = [&](const float* __restrict__ X, int ld, int r0, int R, int k0) -> float4 {
        const int row = r0 + cr, k = k0 + ck;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            const float* px = X + (size_t)row * ld + k;
            // (16-byte path: tile base in SGPRs + 32-bit lane offset, see tile_load16)
            if (VEC) { if (k < g.K) v = tile_load16(X + (size_t)r0 * ld + k0, (unsigned)cr * (unsigned)ld + (unsigned)ck); }
            else {
                if (k + 0 < g.K) v.x = px[0];
                if (k + 1 < g.K) v.y = px[1];
                if (k + 2 < g.K) v.z = px[2];
                if (k + 3 < g.K) v.w = px[3];
            }
        }
        return v;
    };
    auto load_kmajor = [&](int k0) -> float4 {
        const int k = k0 + kr, c = n0 + kc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < g.K) {
            const float* pb = g.B + (size_t)k * g.ldb + c;
            if (VEC) { if (c < g.N) v = tile_load16(g.B + (size_t)k0 * g.ldb + n0, (unsigned)kr * (unsigned)g.ldb + (unsigned)kc); }
            else {
                if (c + 0 < g.N) v.x = pb[0];
                if (c + 1 < g.N) v.y = pb[1];
                if (c + 2 < g.N) v.z = pb[2];
                if (c + 3 < g.N) v.w = pb[3];
            }
        }
        return v;
    };
    auto store_rows = [&](float (*dst)[SLD], const float4& v) {
        dst[ck + 0][cr] = v.x; dst[ck + 1][cr] = v.y; dst[ck + 2][cr] = v.z; dst[ck + 3][cr] = v.w;
    };
    float bv = 0.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
        const int col = n0 + wc * 32 + li;
        bv = col < g.N ? g.bias[col] : 0.f;
    }
    float4 ra = load_rows(g.A, g.lda, m0, g.M, 0);
    float4 rb = TB ? load_rows(g.B, g.ldb, n0, g.N, 0) : load_kmajor(0);
    store_rows(As[0], ra);
    if (TB) store_rows(Bs[0], rb); else *reinterpret_cast<float4*>(&Bs[0][kr][kc]) = rb;
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
        const bool next = k0 + GK < g.K;
        if (next) {
            ra = load_rows(g.A, g.lda, m0, g.M, k0 + GK);
            rb = TB ? load_rows(g.B, g.ldb, n0, g.N, k0 + GK) : load_kmajor(k0 + GK);
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const float a = As[buf][kk + lh][wr * 32 + li];
            const float b = Bs[buf][kk + lh][wc * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (next) {
            store_rows(As[buf ^ 1], ra);
            if (TB) store_rows(Bs[buf ^ 1], rb); else *reinterpret_cast<float4*>(&Bs[buf ^ 1][kr][kc]) = rb;
        }
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = n0 + wc * 32 + li;
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = acc[r];
            if (EPI == EPI_BIAS) v = v + bv;
            if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bv);
            if (row < g.M) g.C[(size_t)(g.rowmap ? g.rowmap[row] : row) * g.ldc + col] = v;
        }
    }
}

#ifdef SERT_VARIANTS
// variants/gemm_stream.h: the projection shapes of a d = 128 model (huge M, N = K = 128) as a streaming kernel --
// measured equal to the tiled kernels (round 4), opt-in through SERT_GEMM_STREAM=1; false when the shape is not its own
template <bool TB, int EPI>
inline bool launch_gemm_stream(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                               int lda, int ldb, int ldc);
#endif

#ifdef SERT_VARIANTS
// variants/gemm_direct.h: A (row-major, contiguous along k) straight from global memory into v_mfma_f32_16x16x4_f32, B in
// 64-k LDS slabs read as ds_read_b128 -- measured equal or slower (round 4); false when the operands are not its own
template <bool TB, int EPI>
inline bool launch_gemm_direct(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                               int lda, int ldb, int ldc);
#endif

// gemm_x3.h: the same contraction on the bf16 matrix pipe (three exact bf16 pieces per operand, six products)
inline bool x3_shape_ok(bool ta, bool tb, const float* A, const float* B, int M, int N, int K, int lda, int ldb, int splits = 1);
template <bool TB, int EPI>
inline void launch_gemm_x3(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                           int lda, int ldb, int ldc, int splits = 1, int kper = 0, size_t c_split_stride = 0);
template <bool CSB>
inline void launch_gemm_x3_ta(hipStream_t s, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                              int ldc, int splits, int kper, size_t c_split_stride);
// SERT_GEMM_FP32=1: every contraction on the fp32 MFMA kernels of this file (cross-check)
// Read ONCE per call into the library (refresh_gemm_choice at the entry points that launch a GEMM: a process can still run
// both ways, tests/test_gpu_fullbatch.py), never between choosing a split count and launching the kernel it was chosen
// for -- the two reads of round 4 could pair x3-sized splits with the fp32 kernel if the variable changed in between.
inline int& gemm_fp32_latch() { static int v = -1; return v; }
inline void refresh_gemm_choice() {
    const char* e = knob("SERT_GEMM_FP32");
    gemm_fp32_latch() = (e && atoi(e) != 0) ? 1 : 0;
}
inline bool gemm_x3_enabled() {
    if (gemm_fp32_latch() < 0) refresh_gemm_choice();
    return gemm_fp32_latch() == 0;
}

// rowmap / mapped_C / mapped (optional): when the launch goes to the 64x64-tile kernel, row r of the product is
// stored as row rowmap[r] of mapped_C (leading dimension ldc) instead of row r of C, and *mapped is set.
template <bool TA, bool TB, int EPI, bool CSB = false>
inline void launch_gemm(hipStream_t s, const float* A, const float* B, float* C, const float* bias,
                        int M, int N, int K, int lda, int ldb, int ldc, int splits = 1,
                        int kper = 0, size_t c_split_stride = 0, unsigned long long* cand = nullptr,
                        unsigned char* cnt = nullptr, int cap = 0, const int32_t* rowmap = nullptr,
                        float* mapped_C = nullptr, bool* mapped = nullptr) {
    if (splits <= 1) { splits = 1; kper = K; }
    if (mapped) *mapped = false;
    if constexpr (TA && !TB && EPI == EPI_STORE) {
        if (!rowmap && (splits == 1 || kper % 16 == 0) && gemm_x3_enabled() && x3_shape_ok(true, false, A, B, M, N, K, lda, ldb, splits)) {
            launch_gemm_x3_ta<CSB>(s, A, B, C, M, N, K, lda, ldb, ldc, splits, kper, splits > 1 ? c_split_stride : 0);
            return;
        }
    }
    if constexpr (!TA && !CSB && (EPI == EPI_STORE || EPI == EPI_BIAS || EPI == EPI_BIAS_TANH)) {
        if ((splits == 1 || (EPI == EPI_STORE && kper % 16 == 0)) && !rowmap && gemm_x3_enabled() &&
            x3_shape_ok(false, TB, A, B, M, N, K, lda, ldb, splits)) {
            launch_gemm_x3<TB, EPI>(s, A, B, C, bias, M, N, K, lda, ldb, ldc, splits, kper, c_split_stride);
            return;
        }
    }
#ifdef SERT_VARIANTS
    if (!TA && !CSB && splits == 1 && (EPI == EPI_STORE || EPI == EPI_BIAS || EPI == EPI_BIAS_TANH)) {
        static const bool stream = variant_knob("SERT_GEMM_STREAM") != nullptr;
        if (stream && launch_gemm_stream<TB, EPI>(s, A, B, C, bias, M, N, K, lda, ldb, ldc)) return;
    }
#endif
#ifdef SERT_VARIANTS
    if (!TA && !CSB && splits == 1 && !rowmap && (EPI == EPI_STORE || EPI == EPI_BIAS || EPI == EPI_BIAS_TANH)) {
        static const int direct_min_k = variant_knob("SERT_GEMM_DIRECT_MIN_K") ? atoi(variant_knob("SERT_GEMM_DIRECT_MIN_K")) : 0;
        if (direct_min_k > 0 && K >= direct_min_k && (long long)M * N >= 128 * 128 * 64 &&
            launch_gemm_direct<TB, EPI>(s, A, B, C, bias, M, N, K, lda, ldb, ldc)) return;
    }
#endif
    GemmArgs g;
    g.rowmap = nullptr;
    // small problem (fewer than two 128x128 tiles per CU): 64x64 tiles, one workgroup each
    static const bool no_small = variant_knob("SERT_GEMM_NO_SMALL") != nullptr;
    // below TWO 128x128 tiles per CU the 64x64 tiles win or draw (round 3 sweep at d = 128: 384 tiles
    // 30.5 -> 26.3 us, the loglinear dG with 347 tiles and K = 1000 162 -> 126 us; 256 and 512 tiles: equal):
    // a CU that gets a second big tile sets the time of the launch, four times as many small ones spread evenly
    static const long long small_below = variant_knob("SERT_GEMM_SMALL_BELOW") ? atoll(variant_knob("SERT_GEMM_SMALL_BELOW")) : 512;   // tuning knob
    // ... and AT two big tiles per CU for a short K (the C2 projections: 512 tiles, K = 128), since the 64x64
    // kernel loads its tiles through buffer loads: 31.3 -> 30.1 and 28.8 -> 27.7 us, C2 step 295.5 -> 292.2 us;
    // at K = 1000 the big tiles keep the boundary (186 against 197 us)
    const long long big_tiles = (long long)cdiv(M, GM) * cdiv(N, GN);
    if (!TA && !CSB && EPI != EPI_FILTER && EPI != EPI_ACCUM && splits == 1 && !no_small &&
        (big_tiles < small_below || (big_tiles == small_below && K <= 512)) && (long long)M * N >= 4 * SM * SM) {
        g.cand = nullptr; g.cnt = nullptr; g.cap = 0;
        g.A = A; g.B = B; g.C = C; g.bias = bias;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
        g.kper = K; g.splits = 1; g.c_split_stride = 0; g.vecA = g.vecB = 0;
        if (rowmap && mapped_C) { g.rowmap = rowmap; g.C = mapped_C; if (mapped) *mapped = true; }
        g.tiles_m = cdiv(M, SM); g.tiles_n = cdiv(N, SM);
        const bool vecs = (lda % 4 == 0) && (ldb % 4 == 0) && lda < (1 << 22) && ldb < (1 << 22) && (((uintptr_t)A) % 16 == 0) &&
                          (((uintptr_t)B) % 16 == 0) && (K % 4 == 0) && (TB ? true : (N % 4 == 0));
        if (vecs) SERT_LAUNCH((gemm_f32_mfma_small<TB, (EPI == EPI_FILTER || EPI == EPI_ACCUM) ? EPI_STORE : EPI, true>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
        else      SERT_LAUNCH((gemm_f32_mfma_small<TB, (EPI == EPI_FILTER || EPI == EPI_ACCUM) ? EPI_STORE : EPI, false>), dim3(g.tiles_m * g.tiles_n), dim3(256), 0, s, g);
        return;
    }
    g.cand = cand; g.cnt = cnt; g.cap = cap;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.kper = kper; g.splits = splits;
    g.tiles_m = cdiv(M, GM); g.tiles_n = cdiv(N, GN);
    g.c_split_stride = c_split_stride;
    g.vecA = g.vecB = 0;
    // vector path: every 16-byte piece aligned and wholly inside or outside
    // (an operand stored k-major -- A of A^T.B, B of A.B -- is contiguous along the tile's M / N axis and its k
    //  remainder is a per-row mask: only an operand that is contiguous along k needs K and kper in whole pieces.
    //  The loglinear dW = G^T.dZ has K = the batch's distinct words, any number: 160 -> 117 us at C2 dims)
    const bool k_pieces = (K % 4 == 0) && (kper % 4 == 0);
    const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && lda < (1 << 22) && ldb < (1 << 22) && (((uintptr_t)A) % 16 == 0) &&
                     (((uintptr_t)B) % 16 == 0) && (k_pieces || (TA && !TB)) &&
                     (TA ? (M % 4 == 0) : true) && (TB ? true : (N % 4 == 0));
    // N just above a multiple of 128 (d = 300): 160-column tiles pad less (gemm_f32_mfma_n160)
    static const bool no_n160 = variant_knob("SERT_GEMM_NO_N160") != nullptr;   // cross-check knob
    if (!no_n160 && vec && EPI != EPI_FILTER && (long long)cdiv(N, GN2) * GN2 * 11 <= (long long)cdiv(N, GN) * GN * 10) {   // >= 10 % less padding
        g.tiles_n = cdiv(N, GN2);
        const long long total160 = (long long)g.tiles_m * g.tiles_n * splits;
        const int grid160 = (int)std::min<long long>(total160, 256 * 3);
        SERT_LAUNCH((gemm_f32_mfma_n160<TA, TB, EPI == EPI_FILTER ? EPI_STORE : EPI, CSB>), dim3(grid160), dim3(256), 0, s, g);
        return;
    }
    const long long total = (long long)g.tiles_m * g.tiles_n * splits;
    // persistent: at most 2 workgroups per CU (256 CUs), each walks items w, w+grid, ...
    static const int max_grid = [] {
        const char* e = variant_knob("SERT_GEMM_GRID");   // tuning knob (default: 2 per CU)
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 256 * SERT_GEMM_WAVES;
    }();
    const int grid = (int)std::min<long long>(total, max_grid);
    const bool full = (M % GM == 0) && (N % GN == 0) && EPI != EPI_FILTER;
    if (vec && full) SERT_LAUNCH((gemm_f32_mfma<TA, TB, EPI, CSB, true, true>), dim3(grid), dim3(256), 0, s, g);
    else if (vec)    SERT_LAUNCH((gemm_f32_mfma<TA, TB, EPI, CSB, true, false>), dim3(grid), dim3(256), 0, s, g);
    else             SERT_LAUNCH((gemm_f32_mfma<TA, TB, EPI, CSB, false, false>), dim3(grid), dim3(256), 0, s, g);
}

// out[i] = sum_s part[s*stride + i] over the split-K partial slabs, in a fixed
// association (G interleaved groups of ascending s, then a fixed G-way add)
// => deterministic.  Elements i < n1 go to out1[i], the rest to out2[i-n1]
// (dW followed by the fused db column sums).  64 elements per workgroup.
template <int G>
__global__ __launch_bounds__(64 * G) void reduce_partials_g(const float* __restrict__ part, int splits,
                                                            size_t stride, size_t count,
                                                            float* __restrict__ out1, size_t n1,
                                                            float* __restrict__ out2,
                                                            const int32_t* __restrict__ rowmap = nullptr, int ncols = 0) {
    // rowmap (optional): out1 is a row-major matrix of `ncols` columns whose row r is stored as row rowmap[r] -- the
    // loglinear dG, whose row u IS the gradient of word uwords[u]: the combine writes it where it belongs instead of a
    // copy kernel behind it (ll_scatter_rows: 5.6 us on the chain of the W3C step)
    __shared__ float red[G][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + l;
    float a = 0.f;
    if (i < count) {
#pragma unroll 8
        for (int s = g; s < splits; s += G) a += part[(size_t)s * stride + i];
    }
    red[g][l] = a;
    __syncthreads();
    if (g == 0 && i < count) {
        float v;
        if (G == 4) {
            v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        } else {
            float q[G / 4];
#pragma unroll
            for (int k = 0; k < G / 4; ++k)
                q[k] = (red[4 * k][l] + red[4 * k + 1][l]) + (red[4 * k + 2][l] + red[4 * k + 3][l]);
            v = q[0];
#pragma unroll
            for (int k = 1; k < G / 4; ++k) v += q[k];
        }
        if (i < n1) {
            if (rowmap) {
                const size_t r = i / (size_t)ncols;
                out1[(size_t)rowmap[r] * ncols + (i - r * ncols)] = v;
            } else out1[i] = v;
        } else out2[i - n1] = v;
    }
}
// Many slabs: sixteen interleaved groups of ascending s per output (1024 threads per 64 outputs)
// -- 512 slabs are 32 loads per thread in four bursts instead of 128 in sixteen (the launch has
// only count / 64 workgroups and is bound by that chain).  The association depends on the slab
// count only (G = 4 below 64 slabs), never on the data.
inline void launch_reduce_partials(hipStream_t s, const float* part, int splits, size_t stride, size_t count,
                                   float* out1, size_t n1, float* out2, const int32_t* rowmap = nullptr, int ncols = 0) {
    const dim3 grid((unsigned)((count + 63) / 64));
    if (splits >= 64)
        hipLaunchKernelGGL((reduce_partials_g<16>), grid, dim3(1024), 0, s, part, splits, stride, count, out1, n1, out2, rowmap, ncols);
    else
        hipLaunchKernelGGL((reduce_partials_g<4>), grid, dim3(256), 0, s, part, splits, stride, count, out1, n1, out2, rowmap, ncols);
}

}  // namespace sert
