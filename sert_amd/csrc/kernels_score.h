// Entity scoring kernels (bin/query.py:239-370, batched), gfx950.
#pragma once
#include "common.h"

namespace sert {

constexpr int kTopKMax = 1024;

// rows /= ||row||_2   (query.py:270-274 entities, :333-336 query projections)
// One wave per row.
__global__ __launch_bounds__(256) void l2_normalize_rows(float* __restrict__ X, int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* x = X + (size_t)r * d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss += x[c] * x[c];
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    for (int c = lane; c < d; c += 64) x[c] = x[c] / nrm;
}

// S <- (S + 1) / 2, in place (query.py:352-357)
__global__ void cos_to_score(float* __restrict__ S, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        S[i] = (S[i] + 1.0f) / 2.0f;
}

// Order-preserving map float -> uint32 such that ascending uint == DESCENDING float.
__device__ __forceinline__ uint32_t desc_key(float f) {
    uint32_t u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending map
    return ~u;                                        // flip => descending
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// Top-k of one row of cosine scores, one workgroup (256 threads) per query:
// 4-pass 8-bit radix select of the k-th largest value, ordered collection
// (ties at the threshold: lowest entity index first), bitonic sort of the k
// survivors by (score desc, index asc).  Emits score = (cos + 1)/2
// (query.py:352-357), computed in fp32.
__global__ __launch_bounds__(256) void topk_rows(const float* __restrict__ S, int V, int k,
                                                 int32_t* __restrict__ idx_out,
                                                 float* __restrict__ val_out) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long keys[kTopKMax];
    __shared__ uint32_t s_prefix, s_krem, s_count, s_ties;
    __shared__ uint32_t wave_cnt[4];
    const int tid = threadIdx.x;
    const float* row = S + (size_t)blockIdx.x * V;

    if (tid == 0) { s_prefix = 0; s_krem = (uint32_t)k; }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const int shift_hi = 8 * (pass + 1);
        for (int e = tid; e < V; e += 256) {
            const uint32_t key = desc_key(row[e]);
            const bool match = (pass == 3) || ((key >> shift_hi) == prefix);
            if (match) atomicAdd(&hist[(key >> (8 * pass)) & 0xffu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t krem = s_krem, b = 0;
            for (; b < 256; ++b) {
                if (hist[b] >= krem) break;
                krem -= hist[b];
            }
            s_krem = krem;
            s_prefix = (prefix << 8) | b;
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;   // key of the k-th best element
    const uint32_t n_ties = s_krem;  // how many elements with key == thr to take
    if (tid == 0) { s_count = 0; s_ties = 0; }
    for (int i = tid; i < kTopKMax; i += 256) keys[i] = ~0ull;
    __syncthreads();
    // elements strictly better than the threshold: any order (sorted below)
    for (int e = tid; e < V; e += 256) {
        const uint32_t key = desc_key(row[e]);
        if (key < thr) {
            const uint32_t pos = atomicAdd(&s_count, 1u);
            keys[pos] = ((unsigned long long)key << 32) | (uint32_t)e;
        }
    }
    __syncthreads();
    // ties at the threshold, in index order
    const uint32_t base = s_count;
    const int lane = tid & 63, wv = tid >> 6;
    for (int e0 = 0; e0 < V; e0 += 256) {
        const int e = e0 + tid;
        const bool is_tie = (e < V) && (desc_key(row[e]) == thr);
        const unsigned long long bal = __ballot(is_tie);
        const uint32_t before_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        uint32_t before = s_ties;
        for (int q = 0; q < wv; ++q) before += wave_cnt[q];
        if (is_tie) {
            const uint32_t t = before + before_in_wave;
            if (t < n_ties) keys[base + t] = ((unsigned long long)thr << 32) | (uint32_t)e;
        }
        __syncthreads();
        if (tid == 0) s_ties += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
        if (s_ties >= n_ties) break;
    }
    __syncthreads();
    // bitonic sort of kTopKMax 64-bit keys ascending (= score desc, index asc)
    for (int size = 2; size <= kTopKMax; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kTopKMax / 2; i += 256) {
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
        idx_out[(size_t)blockIdx.x * k + i] = (int32_t)(uint32_t)kv;
        const float cosv = key_to_float((uint32_t)(kv >> 32));
        val_out[(size_t)blockIdx.x * k + i] = (cosv + 1.0f) / 2.0f;
    }
}

}  // namespace sert
