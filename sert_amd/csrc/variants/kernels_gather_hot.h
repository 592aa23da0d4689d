// The forward's gather with the batch's hot rows staged in LDS: bit-identical, measured 10 % slower (profiles/r05_experiments.txt item 32)
// -- compiled into a -DSERT_VARIANTS build only since round 6.
#pragma once
#include "../kernels_vs.h"

namespace sert {

// ... with the batch's HOT rows in LDS.  A dozen words hold over half of a Zipfian batch's tokens (word_index.h: the dense
// heavy words, <= kHotMax per batch): their rows are staged once per workgroup and a token that points to one reads LDS
// instead of L1 / L2 -- 56 % of C2's 655 k row fetches.  tok_slot[row * n + k] = the hot slot of position k's word or 255.
// The same rows are added in the same (window) order: h is bit for bit vs_gather_mean's.
constexpr int kHotMax = 16;
template <typename IdT>
__global__ __launch_bounds__(256) void vs_gather_mean_hot(const IdT* __restrict__ X, const uint8_t* __restrict__ tok_slot,
                                                          const int32_t* __restrict__ hot_words, int nhot,
                                                          const float* __restrict__ Rw, float* __restrict__ H, int B, int n, int d) {
    extern __shared__ float4 hot_lds[];      // [nhot][d / 4]
    const int chunks = d >> 2;
    for (int i = threadIdx.x; i < nhot * chunks; i += 256) {
        const int h = i / chunks, c = i - h * chunks;
        hot_lds[i] = *reinterpret_cast<const float4*>(Rw + (size_t)hot_words[h] * d + 4 * c);
    }
    __syncthreads();
    const int64_t total = (int64_t)B * chunks;
    const float fn = (float)n;
    for (int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tid < total; tid += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(tid / chunks);
        const int c4 = (int)(tid - (int64_t)row * chunks);
        const IdT* xr = X + (size_t)row * n;
        const uint8_t* sr = tok_slot + (size_t)row * n;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int GC = SERT_GATHER_GC;
        for (int k0 = 0; k0 < n; k0 += GC) {
            size_t id[GC];
            int sl[GC];
#pragma unroll
            for (int q = 0; q < GC; ++q) { id[q] = (size_t)xr[min(k0 + q, n - 1)]; sl[q] = sr[min(k0 + q, n - 1)]; }
            float4 v[GC];
#pragma unroll
            for (int q = 0; q < GC; ++q) {
                if (sl[q] != 255) v[q] = hot_lds[sl[q] * chunks + c4];
                else v[q] = *reinterpret_cast<const float4*>(Rw + id[q] * d + 4 * c4);
            }
#pragma unroll
            for (int q = 0; q < GC; ++q)
                if (k0 + q < n) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        }
        a.x /= fn; a.y /= fn; a.z /= fn; a.w /= fn;
        *reinterpret_cast<float4*>(H + (size_t)row * d + 4 * c4) = a;
    }
}

}  // namespace sert
