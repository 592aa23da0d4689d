// Order-fixed segmented gather-reduce (replaces fp32 atomic scatter-add), gfx950.
#pragma once
#include "common.h"
#include "word_index.h"

namespace sert {

// For every item: acc = sum_{e in [begin,end)} src[row(e), :]  (e ascending),
// then  dst >= 0 : final[dst, :]    = acc / divisor  (gradient table row)
//       dst <  0 : partial[-(dst+1), :] = acc        (next level's input)
// row(e) = rows ? rows[e] / rdiv : e.  LPI lanes cooperate on one item (LPI = 32 when
// d/4 <= 32 so a wave carries two items), each lane owns float4 column chunks.
// DST_SLOT: final rows go to final[item.slot, :] (the word's rank among the batch's
// distinct words) instead of final[item.dst, :] -- loglinear per-distinct-word sums.
// touched (optional): touched[dst] = 1 for every final row written -- the optimiser
// then treats unflagged rows as zero gradient, so the table needs no memset and the
// zeros are never read back.
// LL_FINAL (loglinear distinct-word backward, with DST_SLOT): a final row is stored as
//   mask(lp) * acc - exp(lp) * rsum[slot],  lp = logp[slot, :],  mask = eps <= P <= 1-eps
// (kernels_ll.h: dZu = mask dJsum - P rsum) instead of acc / divisor.
template <int LPI, bool DST_SLOT = false, bool LL_FINAL = false>
__global__ __launch_bounds__(256) void segsum_rows(const float* __restrict__ src,
                                                   const int32_t* __restrict__ rows,
                                                   const int4* __restrict__ items, int nitems,
                                                   float* __restrict__ final_dst,
                                                   float* __restrict__ partial_dst, int d,
                                                   float divisor,
                                                   unsigned char* __restrict__ touched,
                                                   int rdiv = 1,
                                                   const float* __restrict__ logp = nullptr,
                                                   const float* __restrict__ rsum = nullptr) {
    constexpr int IPB = 256 / LPI;  // items per block
    const int sub = threadIdx.x / LPI, l = threadIdx.x % LPI;
    const int item = blockIdx.x * IPB + sub;
    if (item >= nitems) return;
    const int4 it = items[item];
    if (touched && l == 0 && it.z >= 0) touched[it.z] = 1;
    const int chunks = d >> 2;
    // gridDim.y > 1: wide rows (d/4 > LPI) are cut into gridDim.y column groups, one
    // workgroup row each -- more, shorter chains instead of one wave walking the whole row
    // every lane of the LPI group walks the loop (the row numbers travel by lane permute, so no
    // lane may drop out): a lane whose column group is past the row computes on group 0 and
    // stores nothing
    for (int c0 = blockIdx.y * LPI; c0 < chunks; c0 += LPI * gridDim.y) {
        const bool on = c0 + l < chunks;
        const int c = on ? c0 + l : 0;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        // LL_FINAL: the word's log-probabilities and r sum do not depend on the row sums --
        // fetch them now, in the shadow of the row loads, not after them
        float4 lp_pre = make_float4(0.f, 0.f, 0.f, 0.f);
        float rs_pre = 0.f;
        if (LL_FINAL && it.z >= 0) {
            lp_pre = *reinterpret_cast<const float4*>(logp + (size_t)it.w * d + 4 * c);
            rs_pre = rsum[it.w];
        }
        // Row numbers: one coalesced load per LPI entries (lane k holds entry k), handed round
        // with lane permutes -- not a broadcast load per entry in front of every row load.  Rows:
        // eight in flight per trip (then four, then the last one to three), so a 64-entry chunk
        // is 8 + 2 dependent memory round trips instead of 32.  Entries are added left to right
        // whatever the batching, so the sums do not depend on it.
        const int len = it.y - it.x;
        for (int base = 0; base < len; base += LPI) {
            const int cnt = min(LPI, len - base);
            int myr = it.x + base + min(l, cnt - 1);
            if (rows) {
                myr = rows[myr];
                if (rdiv > 1) myr /= rdiv;   // source row = entry / rdiv
            }
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = __shfl(myr, k + q, LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
            if (k + 4 <= cnt) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = __shfl(myr, k + q, LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
                k += 4;
            }
            if (k < cnt) {   // one to three left: the positions past the end repeat the last entry
                float4 v[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int r = __shfl(myr, min(k + q, cnt - 1), LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (k + q < cnt) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
        }
        if (!on) continue;
        if (it.z >= 0 && LL_FINAL) {
            const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
            const size_t o = (size_t)it.w * d + 4 * c;
            const float4 lp = lp_pre;
            const float rs = rs_pre;
            a.x = ((lp.x >= LOGLO && lp.x <= LOGHI) ? a.x : 0.f) - __expf(lp.x) * rs;
            a.y = ((lp.y >= LOGLO && lp.y <= LOGHI) ? a.y : 0.f) - __expf(lp.y) * rs;
            a.z = ((lp.z >= LOGLO && lp.z <= LOGHI) ? a.z : 0.f) - __expf(lp.z) * rs;
            a.w = ((lp.w >= LOGLO && lp.w <= LOGHI) ? a.w : 0.f) - __expf(lp.w) * rs;
            *reinterpret_cast<float4*>(final_dst + o) = a;
        } else if (it.z >= 0) {
            a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
            *reinterpret_cast<float4*>(final_dst + (size_t)(DST_SLOT ? it.w : it.z) * d + 4 * c) = a;
        } else {
            *reinterpret_cast<float4*>(partial_dst + (size_t)(-(it.z + 1)) * d + 4 * c) = a;
        }
    }
}

// Scalar variant for d % 4 != 0 (and for scalars, d = 1): one wave per (item, 64-column group),
// gridDim.y column groups; four entries in flight per trip.  Same summation order as one entry
// at a time (left to right), so the result does not depend on the unrolling.
template <bool DST_SLOT = false>
__global__ __launch_bounds__(256) void segsum_rows_scalar(const float* __restrict__ src,
                                                          const int32_t* __restrict__ rows,
                                                          const int4* __restrict__ items,
                                                          int nitems, float* __restrict__ final_dst,
                                                          float* __restrict__ partial_dst, int d,
                                                          float divisor,
                                                          unsigned char* __restrict__ touched,
                                                          int rdiv = 1) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int4 it = items[item];
    if (touched && lane == 0 && blockIdx.y == 0 && it.z >= 0) touched[it.z] = 1;
    for (int c = blockIdx.y * 64 + lane; c < d; c += 64 * gridDim.y) {
        float a = 0.f;
        int e = it.x;
        for (; e + 4 <= it.y; e += 4) {
            int r0, r1, r2, r3;
            if (rows) { r0 = rows[e]; r1 = rows[e + 1]; r2 = rows[e + 2]; r3 = rows[e + 3]; }
            else      { r0 = e; r1 = e + 1; r2 = e + 2; r3 = e + 3; }
            if (rows && rdiv > 1) { r0 /= rdiv; r1 /= rdiv; r2 /= rdiv; r3 /= rdiv; }
            const float v0 = src[(size_t)r0 * d + c], v1 = src[(size_t)r1 * d + c];
            const float v2 = src[(size_t)r2 * d + c], v3 = src[(size_t)r3 * d + c];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; e < it.y; ++e) {
            const int r = rows ? (rdiv > 1 ? rows[e] / rdiv : rows[e]) : e;
            a += src[(size_t)r * d + c];
        }
        if (it.z >= 0) final_dst[(size_t)(DST_SLOT ? it.w : it.z) * d + c] = a / divisor;
        else partial_dst[(size_t)(-(it.z + 1)) * d + c] = a;
    }
}

// Levels 1 and 2 of a three-level tree in ONE launch (word_index.h: heavy_off).  Workgroups
// [0, nb_normal) take 32 level-1 items each and store the final ones (a chunk item -- dst < 0 -- is
// left to its word's workgroup); workgroup nb_normal + h sums the <= 32 chunk items of heavy word h,
// one lane group each, and adds the chunk sums in chunk order out of LDS: exactly the additions the
// separate level-2 launch makes, in the same order (bit-identical), without its ~6 us of launch.
// (Used when no word of the batch has more than 32 chunk items = 131 072 occurrences; the bench's
// synthetic batches, whose clipped Zipf tail piles 30 % of the tokens on one word, keep the two launches.)
__global__ __launch_bounds__(1024) void segsum_upper_fused(const float* __restrict__ src,
                                                           const int4* __restrict__ items, int nitems,
                                                           int nb_normal, const int4* __restrict__ heavy,
                                                           float* __restrict__ final_dst, int d, float divisor) {
    __shared__ float4 red[64][32];
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int chunks = d >> 2;               // <= 32
    const bool on = l < chunks;
    const int c = on ? l : 0;
    auto sum_rows = [&](int lo, int hi) -> float4 {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = lo;
        for (; e + 8 <= hi; e += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(e + q) * d + 4 * c);
#pragma unroll
            for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        }
        for (; e < hi; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)e * d + 4 * c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        return a;
    };
    if ((int)blockIdx.x < nb_normal) {
        const int item = blockIdx.x * 32 + sub;
        if (item >= nitems) return;
        const int4 it = items[item];
        if (it.z < 0) return;
        float4 a = sum_rows(it.x, it.y);
        a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
        if (on) *reinterpret_cast<float4*>(final_dst + (size_t)it.z * d + 4 * c) = a;
        return;
    }
    const int4 h = heavy[blockIdx.x - nb_normal];     // {first chunk item, chunk items, word, -}
    for (int q = sub; q < h.y; q += 32) {
        const int4 it = items[h.x + q];
        red[q][l] = sum_rows(it.x, it.y);
    }
    __syncthreads();
    if (sub == 0 && on) {
        float4 s = red[0][l];
        for (int q = 1; q < h.y; ++q) {
            const float4 x = red[q][l];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        s.x /= divisor; s.y /= divisor; s.z /= divisor; s.w /= divisor;
        *reinterpret_cast<float4*>(final_dst + (size_t)h.z * d + 4 * c) = s;
    }
}

}  // namespace sert
