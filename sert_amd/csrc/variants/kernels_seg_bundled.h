// Level 0 of the word-gradient tree in bundles of short items: bit-identical, measured slower (round 5) -- compiled into a
// -DSERT_VARIANTS build only since round 6.
#pragma once
#include "../kernels_seg.h"

namespace sert {

// Level 0 of the word-gradient tree, BUNDLED (word_index.h: BatchIndex::bundle_off): a lane group of 32 takes the
// consecutive items [bundles[g], bundles[g + 1]) -- up to eight items, up to kSegChunk entries in all -- and walks their ONE
// consecutive entry range with eight row loads in flight per trip, storing an item's sum when the walk passes its last
// entry.  segsum_rows gives every item a lane group of its own: half of a Zipfian batch's words occur once, so half of the
// lane groups had a single 512-byte load in flight (round 4: 655 k row fetches in 41 us = 8.2 TB/s where the regular window
// gather, ten loads in flight per group, reaches 15.7 TB/s out of the same cache).  Every item is still summed left to right,
// alone, from zero: bit for bit the sums of segsum_rows (tests/test_gpu_parity.py::test_word_gradient_bundled_level0).
// MEASURED SLOWER than segsum_rows (round 5: 59.5 against 56.7 us for the C2 tree) -- opt-in, SERT_SEG_BUNDLE=1.
// d % 4 == 0, d / 4 <= 32 * gridDim.y column groups.
__global__ __launch_bounds__(256) void segsum_rows_bundled(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                           const int4* __restrict__ items, const int32_t* __restrict__ bundles,
                                                           int nbundles, float* __restrict__ final_dst,
                                                           float* __restrict__ partial_dst, int d, float divisor) {
    constexpr int LPI = 32;
    const int sub = threadIdx.x / LPI, l = threadIdx.x % LPI;
    const int g = blockIdx.x * (256 / LPI) + sub;
    if (g >= nbundles) return;
    const int i0 = bundles[g], ni = bundles[g + 1] - i0;
    // lane k < ni holds item i0 + k
    const int4 myit = items[i0 + min(l, ni - 1)];
    const int e_begin = __shfl(myit.x, 0, LPI), e_end = __shfl(myit.y, ni - 1, LPI);
    const int chunks = d >> 2;
    for (int c0 = blockIdx.y * LPI; c0 < chunks; c0 += LPI * gridDim.y) {
        const bool on = c0 + l < chunks;
        const int c = on ? c0 + l : 0;
        int cur = 0;
        int cur_end = __shfl(myit.y, 0, LPI), cur_dst = __shfl(myit.z, 0, LPI);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](const float4 v, int e_next) {
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            if (e_next == cur_end) {       // (uniform over the lane group)
                if (on) {
                    if (cur_dst >= 0) {
                        float4 o = a;
                        o.x /= divisor; o.y /= divisor; o.z /= divisor; o.w /= divisor;
                        *reinterpret_cast<float4*>(final_dst + (size_t)cur_dst * d + 4 * c) = o;
                    } else {
                        *reinterpret_cast<float4*>(partial_dst + (size_t)(-(cur_dst + 1)) * d + 4 * c) = a;
                    }
                }
                a = make_float4(0.f, 0.f, 0.f, 0.f);
                cur = min(cur + 1, ni - 1);
                cur_end = __shfl(myit.y, cur, LPI);
                cur_dst = __shfl(myit.z, cur, LPI);
            }
        };
        for (int base = e_begin; base < e_end; base += LPI) {
            const int cnt = min(LPI, e_end - base);
            const int myr = rows[base + min(l, cnt - 1)];
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = __shfl(myr, k + q, LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) add(v[q], base + k + q + 1);
            }
            if (k < cnt) {   // one to seven left: the positions past the end repeat the last entry and are not added
                float4 v[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int r = __shfl(myr, min(k + q, cnt - 1), LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (k + q < cnt) add(v[q], base + k + q + 1);
            }
        }
    }
}

}  // namespace sert
