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
// relative per operand; products exact in fp32; tests/test_golden_host.py attains 99 % of it).
// With s^_(k) the k-th largest approximate score of a row and T the filter threshold:
//   * every entity with s^ >= T is in the candidate lists (same lists as EPI_FILTER);
//   * at least k candidates have exact score >= s^_(k) - delta, so the exact top k lies among
//     the candidates with s^ >= s^_(k) - 2 delta -- those (~180 of ~600 at k = 100, C5) are
//     re-scored in fp32 by exact_dot32 (one fixed summation order for every path) and sorted;
//   * nothing outside the lists can reach the top k if s^_(k) - delta >= T + delta; a row that
//     fails this is flagged and redone by the materialising fp32 path.
// So the result is exactly the fp32 ranking; bf16 only decides where fp32 is spent.
#pragma once
#include "common.h"
#include "kernels_score.h"

namespace sert {

// 2^-7 + 2^-16 (+ 1e-5 for norms a few ulp above 1) + the fp32 accumulation of d products
// (+ 5e-6: the candidate lists keep the score key without its 6 low bits, <= 64 ulp)
__host__ __device__ inline float bf16_delta(int d) { return 0.00785f + 1.2e-7f * (float)d; }

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
    // four entities per half-wave and trip: their row fetches (random 4 d-byte rows of a table
    // that lives in HBM / Infinity Cache) are in flight together -- one at a time, the loop is
    // a chain of ~1 us load latencies
    for (int i0 = 4 * half; i0 < m; i0 += 32) {
        uint32_t e[4];
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (uint32_t)keys[min(i0 + j, m - 1)];
        if (d == 128) {
            const float4 x = *reinterpret_cast<const float4*>(prow + 4 * l);
            float4 y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = *reinterpret_cast<const float4*>(E + (size_t)e[j] * 128 + 4 * l);
#pragma unroll
            for (int j = 0; j < 4; ++j) {     // same arithmetic as exact_dot32 at d = 128
                float a = fmaf(x.x, y[j].x, 0.f);
                a = fmaf(x.y, y[j].y, a); a = fmaf(x.z, y[j].z, a); a = fmaf(x.w, y[j].w, a);
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                sc[j] = a;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = exact_dot32(prow, E + (size_t)e[j] * d, d, l);
        }
        if (l == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < m) keys[i0 + j] = ((unsigned long long)desc_key(sc[j]) << 32) | e[j];
        }
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

// exclusive prefix sum of one value per thread over a 256-thread workgroup (wave shuffles + 4
// wave totals in LDS: two barriers); *total = sum of all
__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned* wsum /* [4] LDS */, unsigned* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    __syncthreads();                 // previous users of wsum are done
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? wsum[i] : 0u;
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

// topk_from_groups with approximate keys (see the header).  One workgroup per query:
//   gather the row's lists into LDS -> radix-select the k-th best approximate score s^_(k)
//   (4 passes of 8 bits over a 256-bin LDS histogram; a full sort of ~700 keys is not needed)
//   -> keep the entries with s^ >= s^_(k) - 2 delta (order-free LDS append) -> exact fp32
//   re-scoring -> sort those ~2 k entries (score desc, entity asc) -> emit k.
// Flagged for the exact path: overflowed group, fewer than k or more than ccap candidates, or
// no 2 delta gap between s^_(k) and the filter threshold thr[q].  Dynamic LDS: ccap keys.
__global__ __launch_bounds__(256) void topk_from_groups_rescore(
    const uint32_t* __restrict__ cand, const unsigned char* __restrict__ gcnt, int ngroups, int gcap,
    int k, int32_t* __restrict__ idx_out, float* __restrict__ val_out, int q_base, int* __restrict__ nflag,
    int* __restrict__ flag_list, int ccap, const float* __restrict__ P, const float* __restrict__ E, int d,
    const float* __restrict__ thr, float delta) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [ccap]
    __shared__ unsigned hist[256];
    __shared__ unsigned wsum[4];
    __shared__ unsigned s_bad, s_m, s_pos, s_digit, s_rem;
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned char* gc = gcnt + (size_t)q * ngroups;
    // group g = base + tid + 256 i: counts of 8 groups per thread in registers (coalesced byte
    // loads, all in flight together); ngroups > 2048 takes further rounds
    if (tid == 0) { s_bad = 0; s_m = 0; s_pos = 0; }
    unsigned mine = 0;
    bool bad = false;
    for (int base = 0; base < ngroups; base += 2048) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = base + tid + 256 * i;
            const unsigned c = g < ngroups ? gc[g] : 0u;
            bad |= c > (unsigned)gcap;
            mine += c;
        }
    }
    __syncthreads();
    if (bad) s_bad = 1;
    unsigned total;
    (void)block_excl_scan_256(mine, wsum, &total);
    if (s_bad || total < (unsigned)k || total > (unsigned)ccap) {     // workgroup-uniform
        if (tid == 0) flag_list[atomicAdd(nflag, 1)] = q_base + q;
        return;
    }
    // gather (order-free: LDS atomic allocation; selection and the final sort do not depend on
    // the order).  The candidate loads of a thread's 8 groups are issued together, slot by slot.
    for (int base = 0; base < ngroups; base += 2048) {
        unsigned c[8], pos[8];
        unsigned maxc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = base + tid + 256 * i;
            c[i] = g < ngroups ? gc[g] : 0u;
            pos[i] = c[i] ? atomicAdd(&s_pos, c[i]) : 0u;
            maxc = max(maxc, c[i]);
        }
        // entry = score key without its 6 low bits | entity index within the 64-entity group
        const uint32_t* src = cand + ((size_t)q * ngroups + base + tid) * gcap;
        for (unsigned j = 0; j < maxc; ++j) {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (j < c[i]) v[i] = src[(size_t)256 * i * gcap + j];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (j < c[i])
                    keys[pos[i] + j] = ((unsigned long long)(v[i] & ~63u) << 32) |
                                       (unsigned)((base + tid + 256 * i) * 64 + (int)(v[i] & 63u));
        }
    }
    // radix select on the descending score keys: the k-th smallest high word
    uint32_t prefix = 0;
    unsigned rem = (unsigned)k;                 // rank still to find among the entries matching prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();                        // (also orders the gather before the first pass)
        const uint32_t pmask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        for (int i = tid; i < (int)total; i += 256) {
            const uint32_t h = (uint32_t)(keys[i] >> 32);
            if ((h & pmask) == prefix) atomicAdd(&hist[(h >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        const unsigned c = hist[tid];
        unsigned dummy;
        const unsigned before = block_excl_scan_256(c, wsum, &dummy);
        if (before < rem && rem <= before + c) { s_digit = (unsigned)tid; s_rem = rem - before; }
        __syncthreads();
        prefix |= s_digit << shift;
        rem = s_rem;
    }
    const float sk = key_to_float(prefix);      // s^_(k)
    if (!(sk - delta >= thr[q] + delta)) {      // workgroup-uniform
        if (tid == 0) flag_list[atomicAdd(nflag, 1)] = q_base + q;
        return;
    }
    const uint32_t cut = desc_key(sk - 2.0f * delta);
    // keep the entries at or above the cut, compacted in place: every thread first reads its
    // strided share (ccap <= 4096 -> at most 16 entries), then all write after a barrier
    unsigned long long kv[16];
    unsigned keepmask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        kv[i] = e < (int)total ? keys[e] : ~0ull;
        if ((uint32_t)(kv[i] >> 32) <= cut) keepmask |= 1u << i;     // (~0 never passes: cut < 2^32 - 1)
    }
    __syncthreads();
    unsigned o = keepmask ? atomicAdd(&s_m, (unsigned)__popc(keepmask)) : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((keepmask >> i) & 1u) keys[o++] = kv[i];
    __syncthreads();
    const int m = (int)s_m;
    int sn = 2;
    while (sn < m) sn <<= 1;
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
    const uint16_t* E16;   // (N, kp) bf16, row stride estride elements
    int M, N, kp;          // kp % 32 == 0
    size_t estride;
    float* C; int ldc;     // STORE: fp32 scores (the 1/16 sample that sets the thresholds)
    int tiles_m, tiles_n;
    const float* thr;
    uint32_t* cand;             // [M][ngr][cap]: score key & ~63 | entity & 63
    unsigned char* cnt;         // [M][ngr], zeroed by the caller
    int ngr, cap;
};

// The filtering epilogue of one 128 x 128 tile (cf. gemm.h EPI_FILTER), shared by the two filter kernels below.
// acc: the wave's two 32 x 32 blocks (columns wc * 64 + {0, 32} + li of the tile).
__device__ __forceinline__ void score_filter_epilogue(f32x16_t (&acc)[2], const ScoreBf16Args& g, int tn, int m0, int n0, int wr,
                                                      int wc, int li, int lh, const float* thr_s) {
    const int nrem = g.N - n0;
    if (nrem < SB_T) {       // edge tile: the (clamped, duplicated) columns beyond N never pass
        const int col0e = wc * 64 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (col0e >= nrem) acc[0][r] = -INFINITY;
            if (col0e + 32 >= nrem) acc[1][r] = -INFINITY;
        }
    }
    // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): the 32 lanes of a
    // half-wave hold 32 consecutive columns of one row, so the slot of an element in its
    // (row, 64-column group) list is a ballot/popcount prefix over the group's two halves --
    // no atomics, ascending column order.  VALU-bound kernel: the no-candidate case is two
    // compares and a branch, the candidate case one divergent region with selects.
    const unsigned below = (1u << li) - 1u;
    const unsigned ngr = (unsigned)g.ngr, ucap = (unsigned)g.cap;
    const size_t gbase = (size_t)m0 * ngr + 2u * (unsigned)tn;
    uint32_t* cand_t = g.cand + gbase * ucap;
    unsigned char* cnt_t = g.cnt + gbase;
    const int row0 = wr * 32 + 4 * lh;
    float th[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) th[r] = thr_s[row0 + (r & 3) + 8 * (r >> 2)];
    unsigned goff = (unsigned)row0 * ngr + (unsigned)wc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v0 = acc[0][r], v1 = acc[1][r];
        const bool p0 = v0 >= th[r], p1 = v1 >= th[r];
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(p0), b1 = __builtin_amdgcn_ballot_w64(p1);
        if (b0 | b1) {                                   // wave-uniform
            const unsigned h0 = lh ? (unsigned)(b0 >> 32) : (unsigned)b0;
            const unsigned h1 = lh ? (unsigned)(b1 >> 32) : (unsigned)b1;
            if (p0 | p1) {
                const unsigned n0c = __popc(h0);
                const unsigned slot = p0 ? __popc(h0 & below) : n0c + __popc(h1 & below);
                const float v = p0 ? v0 : v1;
                const unsigned cl = (unsigned)li + (p0 ? 0u : 32u);
                const unsigned at = goff * ucap;
                if (slot < ucap) cand_t[at + slot] = (desc_key(v) & ~63u) | cl;
                if (slot == 0) {                          // exactly one lane per non-empty list
                    const unsigned tot = n0c + __popc(h1);
                    cnt_t[goff] = (unsigned char)(tot > 255u ? 255u : tot);
                }
                if (p0 & p1) {                            // both halves of this lane's pair (rare)
                    const unsigned s1 = n0c + __popc(h1 & below);
                    if (s1 < ucap) cand_t[at + s1] = (desc_key(v1) & ~63u) | ((unsigned)li + 32u);
                }
            }
        }
        goff += ((r & 3) == 3) ? 5u * ngr : ngr;
    }
}

// The same lists, slots handed out by LDS atomics (round 5).  The consumer (topk_from_groups_rescore) gathers the lists
// order-free and flags an overflowed group for the exact path, so the ORDER of a list's entries is free -- and with it the
// ballot / popcount ranking above, which is what this kernel spends its time on (VALU-bound: ~480 epilogue instructions per
// wave and tile against 16 MFMAs; half of the sixteen row iterations find a candidate at the 0.6 % density the sampled threshold
// gives).  Here a lane that holds a candidate takes its slot from the row's counter (cnt_w: this wave's 32 rows, one
// 64-column group each) and stores; the counts are written once at the end, for all 32 rows (zero included).
__device__ __forceinline__ void score_filter_epilogue_atomic(f32x16_t (&acc)[2], const ScoreBf16Args& g, int tn, int m0, int n0, int wr,
                                                             int wc, int li, int lh, const float* thr_s, unsigned* cnt_w) {
    const int nrem = g.N - n0;
    if (nrem < SB_T) {       // edge tile: the (clamped, duplicated) columns beyond N never pass
        const int col0e = wc * 64 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (col0e >= nrem) acc[0][r] = -INFINITY;
            if (col0e + 32 >= nrem) acc[1][r] = -INFINITY;
        }
    }
    const int lane = li + 32 * lh;
    if (lane < 32) cnt_w[lane] = 0u;            // (same wave writes and reads: LDS operations of a wave complete in order)
    const unsigned ngr = (unsigned)g.ngr, ucap = (unsigned)g.cap;
    const size_t gbase = (size_t)m0 * ngr + 2u * (unsigned)tn;
    uint32_t* cand_t = g.cand + gbase * ucap;
    unsigned char* cnt_t = g.cnt + gbase;
    const int row0 = wr * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * lh;              // row inside the wave's 32
        const float th = thr_s[wr * 32 + rl];
        const float v0 = acc[0][r], v1 = acc[1][r];
        const bool p0 = v0 >= th, p1 = v1 >= th;
        if (__builtin_amdgcn_ballot_w64(p0 | p1)) {                  // wave-uniform
            const unsigned goff = (unsigned)(wr * 32 + rl) * ngr + (unsigned)wc;
            if (p0) {
                const unsigned slot = atomicAdd(&cnt_w[rl], 1u);
                if (slot < ucap) cand_t[(size_t)goff * ucap + slot] = (desc_key(v0) & ~63u) | (unsigned)li;
            }
            if (p1) {
                const unsigned slot = atomicAdd(&cnt_w[rl], 1u);
                if (slot < ucap) cand_t[(size_t)goff * ucap + slot] = (desc_key(v1) & ~63u) | ((unsigned)li + 32u);
            }
        }
    }
    (void)row0;
    if (lane < 32) {
        const unsigned tot = cnt_w[lane];
        if (m0 + wr * 32 + lane < g.M)                               // (rows past M do not exist: no list, no count)
            cnt_t[(unsigned)(wr * 32 + lane) * ngr + (unsigned)wc] = (unsigned char)(tot > 255u ? 255u : tot);
    }
}

template <bool STORE, bool ATOMIC_EPI = false>
__global__ __launch_bounds__(512, 4) void score_filter_bf16(const ScoreBf16Args g) {
    __shared__ __attribute__((aligned(16))) unsigned char As[SB_T * SB_LDB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[SB_T * SB_LDB];
    __shared__ float thr_s[SB_T];
    __shared__ unsigned cnt_s[8][32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;            // 4 x 2 waves
    const int li = lane & 31, lh = lane >> 5;
    // tile index runs along M first: the workgroups in flight share a few entity tiles
    const int tn = blockIdx.x / g.tiles_m, tm = blockIdx.x - tn * g.tiles_m;
    const int m0 = tm * SB_T, n0 = tn * SB_T;
    if (!STORE && tid < SB_T) thr_s[tid] = m0 + tid < g.M ? g.thr[m0 + tid] : INFINITY;

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
            const unsigned char* pb = (const unsigned char*)g.E16 + (size_t)min(n0 + r, g.N - 1) * g.estride * 2 + (size_t)(kc + lkq * 8) * 2;
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
    if (STORE) {
        const int mrem = g.M - m0;
        float* Ct = g.C + (size_t)m0 * g.ldc + n0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = wc * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                if (row < mrem && col < nrem) Ct[(size_t)row * g.ldc + col] = acc[b][r];
            }
        }
        return;
    }
    if (ATOMIC_EPI) score_filter_epilogue_atomic(acc, g, tn, m0, n0, wr, wc, li, lh, thr_s, cnt_s[w]);
    else score_filter_epilogue(acc, g, tn, m0, n0, wr, wc, li, lh, thr_s);
}

#ifdef SERT_VARIANTS
__global__ void score_filter_bf16_ring(const ScoreBf16Args g, int tiles_per_wg);
#endif

inline void launch_score_filter_bf16(hipStream_t s, const uint16_t* P16, const uint16_t* E16, const float* thr,
                                     uint32_t* cand, unsigned char* cnt, int ngr, int cap, int M,
                                     int N, int kp) {
    ScoreBf16Args g = {};
    g.P16 = P16; g.E16 = E16; g.M = M; g.N = N; g.kp = kp; g.estride = (size_t)kp;
    g.tiles_m = cdiv(M, SB_T); g.tiles_n = cdiv(N, SB_T);
    g.thr = thr; g.cand = cand; g.cnt = cnt; g.ngr = ngr; g.cap = cap;
#ifdef SERT_VARIANTS
    // persistent form with the queries' fragments in registers and the entity tiles by LDS-DMA (variants/score_filter_ring.h):
    // exact, 1.59 ms per 10 000 queries against 1.29 -- the kernel is bound by its epilogue, not by its loads
    static const bool ring = variant_knob("SERT_SCORE_RING") != nullptr;
    if (kp == 128 && ring) {
        const int parts = std::max(1, std::min(g.tiles_n, cdiv(2 * 256, g.tiles_m)));
        const int per = cdiv(g.tiles_n, parts);
        hipLaunchKernelGGL(score_filter_bf16_ring, dim3(g.tiles_m * cdiv(g.tiles_n, per)), dim3(512), 0, s, g, per);
        return;
    }
#endif
    // (the epilogue with slots from LDS atomics, score_filter_epilogue_atomic: measured EQUAL -- 1.221 / 1.261 ms per 10 000
    //  queries against 1.235 / 1.210, identical results -- so the kernel is not bound by its epilogue's instruction count
    //  after all; variants build only, SERT_SCORE_EPI=atomic; profiles/r05_experiments.txt)
    static const bool atomic_epi = variant_knob("SERT_SCORE_EPI") && !strcmp(variant_knob("SERT_SCORE_EPI"), "atomic");
    if (atomic_epi) hipLaunchKernelGGL((score_filter_bf16<false, true>), dim3(g.tiles_m * g.tiles_n), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((score_filter_bf16<false, false>), dim3(g.tiles_m * g.tiles_n), dim3(512), 0, s, g);
}

// C (M, N) = P16 . E16[::stride]^T in fp32 (approximate scores of every stride-th entity)
inline void launch_score_sample_bf16(hipStream_t s, const uint16_t* P16, const uint16_t* E16, float* C, int M,
                                     int N, int kp, int stride) {
    ScoreBf16Args g = {};
    g.P16 = P16; g.E16 = E16; g.M = M; g.N = N; g.kp = kp; g.estride = (size_t)kp * stride;
    g.tiles_m = cdiv(M, SB_T); g.tiles_n = cdiv(N, SB_T);
    g.C = C; g.ldc = N;
    hipLaunchKernelGGL((score_filter_bf16<true, false>), dim3(g.tiles_m * g.tiles_n), dim3(512), 0, s, g);
}

}  // namespace sert
