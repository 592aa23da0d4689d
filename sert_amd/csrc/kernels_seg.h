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
    for (int c = blockIdx.y * LPI + l; c < chunks; c += LPI * gridDim.y) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        // LL_FINAL: the word's log-probabilities and r sum do not depend on the row sums --
        // fetch them now, in the shadow of the row loads, not after them
        float4 lp_pre = make_float4(0.f, 0.f, 0.f, 0.f);
        float rs_pre = 0.f;
        if (LL_FINAL && it.z >= 0) {
            lp_pre = *reinterpret_cast<const float4*>(logp + (size_t)it.w * d + 4 * c);
            rs_pre = rsum[it.w];
        }
        int e = it.x;
        for (; e + 4 <= it.y; e += 4) {
            int r0, r1, r2, r3;
            if (rows) { r0 = rows[e]; r1 = rows[e + 1]; r2 = rows[e + 2]; r3 = rows[e + 3]; }
            else      { r0 = e; r1 = e + 1; r2 = e + 2; r3 = e + 3; }
            if (rows && rdiv > 1) { r0 /= rdiv; r1 /= rdiv; r2 /= rdiv; r3 /= rdiv; }   // source row = entry / rdiv
            const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)r0 * d + 4 * c);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)r1 * d + 4 * c);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)r2 * d + 4 * c);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)r3 * d + 4 * c);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; e < it.y; ++e) {
            const int r = rows ? (rdiv > 1 ? rows[e] / rdiv : rows[e]) : e;
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
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

}  // namespace sert
