// Query scoring, bf16 prefilter + exact fp32 rescoring (gfx950).
//
// bin/query.py:288-365 ranks entities by the fp32 cosine.  The fused path (kernels_score.h)
// spends 90 % of its time in an fp32 MFMA GEMM whose only job is to find the ~600 of 100 000
// entities per query that reach a threshold -- a job that does not need fp32.  Here that GEMM
// runs on the bf16 matrix pipe (16x the fp32 MFMA rate) over bf16 copies of the unit-norm
// operands, and fp32 enters only for the few entities that can still be in the top k:
//
//   |s^ - s| <= sum_i |a_i b_i| ((1 + 2^-8)^2 - 1) + fp32 accumulation  <=  delta = bf16_delta(d)
//
// for vectors of norm <= 1 (bf16 keeps 8 significant bits: round-to-nearest error <= 2^-8
// relative per operand; products exact in fp32; tests/test_golden_host.py attains 99 % of it).  With s^_(k) the k-th largest approximate score of a row and T the filter threshold:
//   * every entity with s^ >= T is in the candidate lists (same lists as EPI_FILTER);
//   * at least k candidates have exact score >= s^_(k) - delta, so the exact top k lies among
//     the candidates with s^ >= s^_(k) - 2 delta -- those (~2 k of them) are re-scored in
//     fp32 by exact_dot (one fixed summation order for every path) and sorted;
//   * nothing outside the lists can reach the top k if s^_(k) - delta >= T + delta; a row that
//     fails this is flagged and redone by the materialising fp32 path.
// So the result is exactly the fp32 ranking; bf16 only decides where fp32 is spent.
#pragma once
#include "common.h"
#include "kernels_score.h"

namespace sert {

// 2^-7 + 2^-16 (+ 1e-5 for norms a few ulp above 1) + the fp32 accumulation of d products
__host__ __device__ inline float bf16_delta(int d) { return 0.00784f + 1.2e-7f * (float)d; }

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// fp32 (rows, d) -> bf16 (rows, kp), round to nearest even, zero padded to kp columns
__global__ void to_bf16_rows(const float* __restrict__ src, int64_t rows, int d, int kp,
                             uint16_t* __restrict__ dst) {
    const size_t n = (size_t)rows * kp;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / kp;
        const int c = (int)(t - r * kp);
        uint32_t u = c < d ? __float_as_uint(src[r * d + c]) : 0u;
        u += 0x7fffu + ((u >> 16) & 1u);
        dst[t] = (uint16_t)(u >> 16);
    }
}

// The one fp32 dot product every reported score comes from: lanes of a 32-lane half-wave own
// float4 chunks 4 l, 4 l + 128, ...; fmaf chain per lane, then a fixed xor-shuffle tree.
// All 32 lanes of the half return the sum.
__device__ __forceinline__ float exact_dot32(const float* __restrict__ p, const float* __restrict__ e, int d, int l) {
    float a = 0.f;
    for (int c = 4 * l; c < d; c += 128) {
        const float4 x = *reinterpret_cast<const float4*>(p + c);
        const float4 y = *reinterpret_cast<const float4*>(e + c);
        a = fmaf(x.x, y.x, a); a = fmaf(x.y, y.y, a); a = fmaf(x.z, y.z, a); a = fmaf(x.w, y.w, a);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    return a;
}

// keys[0, m): (anything << 32 | entity).  Replace the high words by the exact fp32 score keys,
// sort (score desc, entity asc) and emit the k best.  sort_n = power of two >= m, <= LDS size.
__device__ __forceinline__ void rescore_sort_emit(unsigned long long* keys, int m, int sort_n,
                                                  const float* __restrict__ prow, const float* __restrict__ E,
                                                  int d, int k, int32_t* __restrict__ idx_out,
                                                  float* __restrict__ val_out) {
    const int tid = threadIdx.x, half = tid >> 5, l = tid & 31;
    for (int i = half; i < m; i += 8) {
        const uint32_t e = (uint32_t)keys[i];
        const float s = exact_dot32(prow, E + (size_t)e * d, d, l);
        if (l == 0) keys[i] = ((unsigned long long)desc_key(s) << 32) | e;
    }
    for (int i = m + tid; i < sort_n; i += 256) keys[i] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= sort_n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < sort_n / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const unsigned long long kv = keys[i];
        idx_out[i] = (int32_t)(uint32_t)kv;
        val_out[i] = (key_to_float((uint32_t)(kv >> 32)) + 1.0f) / 2.0f;
    }
}

// Materialising path under the bf16 scorer: its k winners per row (picked on fp32 GEMM values)
// get the same exact_dot scores and ordering as the fused path's.
__global__ __launch_bounds__(256) void rescore_topk_rows(const float* __restrict__ P, const float* __restrict__ E,
                                                         int d, int k, int32_t* __restrict__ idx,
                                                         float* __restrict__ val) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += 256) keys[i] = (uint32_t)idx[(size_t)q * k + i];
    __syncthreads();
    int sort_n = 2;
    while (sort_n < k) sort_n <<= 1;
    rescore_sort_emit(keys, k, sort_n, P + (size_t)q * d, E, d, k, idx + (size_t)q * k, val + (size_t)q * k);
}

// topk_from_groups with approximate keys: gather + sort the row's lists as before, then cut at
// s^_(k) - 2 delta, re-score, re-sort (see the header).  thr = the filter thresholds T.
__global__ __launch_bounds__(256) void topk_from_groups_rescore(
    const unsigned long long* __restrict__ cand, const unsigned char* __restrict__ gcnt, int ngroups, int gcap,
    int k, int32_t* __restrict__ idx_out, float* __restrict__ val_out, int q_base, int* __restrict__ nflag,
    int* __restrict__ flag_list, int ccap, const float* __restrict__ P, const float* __restrict__ E, int d,
    const float* __restrict__ thr, float delta) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    __shared__ unsigned scan[256];
    __shared__ unsigned s_bad, s_m;
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned char* gc = gcnt + (size_t)q * ngroups;
    const int per = (ngroups + 255) / 256;
    const int g0 = tid * per, g1 = min(ngroups, g0 + per);
    unsigned mine = 0;
    bool bad = false;
    for (int g = g0; g < g1; ++g) {
        const unsigned c = gc[g];
        bad |= c > (unsigned)gcap;
        mine += c;
    }
    if (tid == 0) { s_bad = 0; s_m = 0; }
    scan[tid] = mine;
    __syncthreads();
    if (bad) s_bad = 1;
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned v = (tid >= off) ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const unsigned total = scan[255];
    if (s_bad || total < (unsigned)k || total > (unsigned)ccap) {     // workgroup-uniform
        if (tid == 0) flag_list[atomicAdd(nflag, 1)] = q_base + q;
        return;
    }
    unsigned pos = scan[tid] - mine;
    for (int g = g0; g < g1; ++g) {
        const unsigned c = gc[g];
        const unsigned long long* src = cand + ((size_t)q * ngroups + g) * gcap;
        for (unsigned j = 0; j < c; ++j) keys[pos + j] = src[j];
        pos += c;
    }
    int sort_n = 2;
    while (sort_n < (int)total) sort_n <<= 1;
    for (int i = (int)total + tid; i < sort_n; i += 256) keys[i] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= sort_n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < sort_n / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    // s^_(k) and the two conditions of the header
    const float sk = key_to_float((uint32_t)(keys[k - 1] >> 32));
    if (!(sk - delta >= thr[q] + delta)) {                  // workgroup-uniform
        if (tid == 0) flag_list[atomicAdd(nflag, 1)] = q_base + q;
        return;
    }
    const uint32_t cut = desc_key(sk - 2.0f * delta);            // keep keys <= cut (descending keys)
    unsigned cntm = 0;
    for (int i = tid; i < (int)total; i += 256) cntm += ((uint32_t)(keys[i] >> 32) <= cut) ? 1u : 0u;
    if (cntm) atomicAdd(&s_m, cntm);
    __syncthreads();
    const int m = (int)s_m;            // sorted descending: exactly the first m entries
    int sn = 2;
    while (sn < m) sn <<= 1;
    __syncthreads();
    rescore_sort_emit(keys, m, sn, P + (size_t)q * d, E, d, k, idx_out + (size_t)q * k, val_out + (size_t)q * k);
}

// ---- the bf16 filter GEMM ---------------------------------------------------------------
// Workgroup tile 128 queries x 128 entities, 8 waves of 32x64 (1x2 v_mfma_f32_32x32x16_bf16
// blocks), K in chunks of 64 staged in LDS (37 KB: four workgroups per CU).  The kernel is epilogue-bound (32 MFMA-cycles per
// element-lane against a compare/ballot/rank epilogue), so the shape is chosen for resident
// waves -- four workgroups = 32 waves per CU -- not for MFMA efficiency.
// Epilogue = EPI_FILTER of gemm.h (same C layout, same lists).
constexpr int SB_T = 128, SB_KC = 64, SB_LDB = 2 * SB_KC + 16;   // bytes per LDS row (144: conflict-free 16-byte reads)

struct ScoreBf16Args {
    const uint16_t* P16;   // (M, kp) bf16
    const uint16_t* E16;   // (N, kp) bf16
    int M, N, kp;          // kp % 32 == 0
    int tiles_m, tiles_n;
    const float* thr;
    unsigned long long* cand;   // [M][ngr][cap]
    unsigned char* cnt;         // [M][ngr], zeroed by the caller
    int ngr, cap;
};

__global__ __launch_bounds__(512, 4) void score_filter_bf16(const ScoreBf16Args g) {
    __shared__ __attribute__((aligned(16))) unsigned char As[SB_T * SB_LDB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[SB_T * SB_LDB];
    __shared__ float thr_s[SB_T];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;            // 4 x 2 waves
    const int li = lane & 31, lh = lane >> 5;
    // tile index runs along M first: the workgroups in flight share a few entity tiles
    const int tn = blockIdx.x / g.tiles_m, tm = blockIdx.x - tn * g.tiles_m;
    const int m0 = tm * SB_T, n0 = tn * SB_T;
    if (tid < SB_T) thr_s[tid] = m0 + tid < g.M ? g.thr[m0 + tid] : INFINITY;

    f32x16_t acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // loader: 16-byte pieces; piece p = tid + 512 i -> row p / 8, k-piece p % 8 (a row's 8
    // pieces are read by 8 consecutive lanes: 128 contiguous bytes)
    const int lrow = tid >> 3, lkq = tid & 7;
    const size_t rowb = (size_t)g.kp * 2;
    for (int kc = 0; kc < g.kp; kc += SB_KC) {
        const int kw = min(SB_KC, g.kp - kc);          // multiple of 32
        uint4 ra[2], rb[2];
        const bool kin = lkq * 8 < kw;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + 64 * i;
            const unsigned char* pa = (const unsigned char*)g.P16 + (size_t)min(m0 + r, g.M - 1) * rowb + (size_t)(kc + lkq * 8) * 2;
            const unsigned char* pb = (const unsigned char*)g.E16 + (size_t)min(n0 + r, g.N - 1) * rowb + (size_t)(kc + lkq * 8) * 2;
            ra[i] = kin ? *reinterpret_cast<const uint4*>(pa) : make_uint4(0, 0, 0, 0);
            rb[i] = kin ? *reinterpret_cast<const uint4*>(pb) : make_uint4(0, 0, 0, 0);
        }
        if (kc) __syncthreads();                       // previous chunk's fragments all read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + 64 * i;
            *reinterpret_cast<uint4*>(As + r * SB_LDB + lkq * 16) = ra[i];
            *reinterpret_cast<uint4*>(Bs + r * SB_LDB + lkq * 16) = rb[i];
        }
        __syncthreads();
        // fragment of a 32x16 block: lane -> row li, k = 8 lh .. 8 lh + 7 (16 bytes)
        for (int ks = 0; ks < kw; ks += 16) {
            const int ko = (ks + 8 * lh) * 2;
            const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + (wr * 32 + li) * SB_LDB + ko);
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bs + (wc * 64 + li) * SB_LDB + ko);
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bs + (wc * 64 + 32 + li) * SB_LDB + ko);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
        }
    }

    // ---- filtering epilogue (cf. gemm.h EPI_FILTER) ----
    // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The 32 lanes of
    // a half-wave hold 32 consecutive columns of one row: slot = ballot/popcount prefix over the
    // two 32-column halves of the row's 64-column group.  No atomics, deterministic order.
    const int nrem = g.N - n0;
    const unsigned below = (1u << li) - 1u;
    const unsigned ngr = (unsigned)g.ngr, ucap = (unsigned)g.cap;
    const size_t gbase = (size_t)m0 * ngr + 2u * (unsigned)tn;
    unsigned long long* cand_t = g.cand + gbase * ucap;
    unsigned char* cnt_t = g.cnt + gbase;
    const int row0 = wr * 32 + 4 * lh;
    const int col0 = wc * 64 + li, col1 = col0 + 32;
    const bool c0ok = col0 < nrem, c1ok = col1 < nrem;
    float th[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) th[r] = thr_s[row0 + (r & 3) + 8 * (r >> 2)];
    unsigned goff = (unsigned)row0 * ngr + (unsigned)wc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v0 = acc[0][r], v1 = acc[1][r];
        const bool p0 = c0ok && v0 >= th[r];
        const bool p1 = c1ok && v1 >= th[r];
        const unsigned h0 = (unsigned)(__builtin_amdgcn_ballot_w64(p0) >> (32 * lh));
        const unsigned h1 = (unsigned)(__builtin_amdgcn_ballot_w64(p1) >> (32 * lh));
        if (h0 | h1) {
            const unsigned n0c = __popc(h0);
            if (p0) {
                const unsigned slot = __popc(h0 & below);
                if (slot < ucap)
                    cand_t[(size_t)goff * ucap + slot] = ((unsigned long long)desc_key(v0) << 32) | (unsigned)(n0 + col0);
            }
            if (p1) {
                const unsigned slot = n0c + __popc(h1 & below);
                if (slot < ucap)
                    cand_t[(size_t)goff * ucap + slot] = ((unsigned long long)desc_key(v1) << 32) | (unsigned)(n0 + col1);
            }
            if (li == 0) {
                const unsigned tot = n0c + __popc(h1);
                cnt_t[goff] = (unsigned char)(tot > 255u ? 255u : tot);
            }
        }
        goff += ((r & 3) == 3) ? 5u * ngr : ngr;
    }
}

inline void launch_score_filter_bf16(hipStream_t s, const uint16_t* P16, const uint16_t* E16, const float* thr,
                                     unsigned long long* cand, unsigned char* cnt, int ngr, int cap, int M,
                                     int N, int kp) {
    ScoreBf16Args g;
    g.P16 = P16; g.E16 = E16; g.M = M; g.N = N; g.kp = kp;
    g.tiles_m = cdiv(M, SB_T); g.tiles_n = cdiv(N, SB_T);
    g.thr = thr; g.cand = cand; g.cnt = cnt; g.ngr = ngr; g.cap = cap;
    hipLaunchKernelGGL(score_filter_bf16, dim3(g.tiles_m * g.tiles_n), dim3(512), 0, s, g);
}

}  // namespace sert
