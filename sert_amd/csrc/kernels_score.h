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

// Visit every element of a row with 16-byte loads, 4 independent loads per
// thread in flight (a 4-byte load per iteration left this kernel latency-bound:
// 2.8 TB/s with 32 waves per CU).  f(value, index) is called in arbitrary order.
template <typename F>
__device__ __forceinline__ void topk_scan_row(const float* __restrict__ row, int V, F f) {
    const int tid = threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(row)) & 15) == 0) {
        const int V4 = V >> 2;
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int i0 = tid; i0 < V4; i0 += 4 * 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < V4) v[u] = r4[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < V4) {
                    f(v[u].x, 4 * i); f(v[u].y, 4 * i + 1); f(v[u].z, 4 * i + 2); f(v[u].w, 4 * i + 3);
                }
            }
        }
        for (int e = (V4 << 2) + tid; e < V; e += 256) f(row[e], e);
    } else {
        for (int e = tid; e < V; e += 256) f(row[e], e);
    }
}

constexpr int kTopKCand = 2048;   // LDS capacity for the threshold-bin candidates

// Top-k of one row of cosine scores, one workgroup (256 threads) per query.
//   pass 1 (1 read)  2048-bin histogram of the top 11 bits of the order-preserving
//                    key -> the bin holding the k-th best score
//   pass 2 (1 read)  scores in better bins are winners; the threshold bin's
//                    elements go to an LDS candidate list
//   in LDS           bitonic sort of the candidates by (score desc, index asc),
//                    the best `need` join the winners; final sort of the k winners
// Two reads of the row instead of six for the plain 4x8-bit radix select, which
// remains as the fallback when the threshold bin overflows the candidate list
// (many equal scores).  Ties resolve to the lowest entity index.  Emits
// score = (cos + 1)/2 (query.py:352-357), computed in fp32.
// thr_out (optional): instead of the (index, score) lists, only the RAW cosine of the
// k-th best element is written per row (the sampled threshold of the fused path).
__global__ __launch_bounds__(256) void topk_rows(const float* __restrict__ S, int V, int k,
                                                 int32_t* __restrict__ idx_out,
                                                 float* __restrict__ val_out,
                                                 float* __restrict__ thr_out = nullptr) {
    __shared__ uint32_t hist[2048];
    __shared__ unsigned long long keys[kTopKMax];
    __shared__ unsigned long long cand[kTopKCand];
    __shared__ uint32_t scan_tmp[256];
    __shared__ uint32_t s_bin, s_need, s_count, s_ncand, s_prefix, s_krem, s_ties;
    __shared__ uint32_t wave_cnt[4];
    const int tid = threadIdx.x;
    const float* row = S + (size_t)blockIdx.x * V;
    int sort_n = 2;
    while (sort_n < k) sort_n <<= 1;

    // ---- pass 1: 11-bit histogram ---------------------------------------------
    for (int b = tid; b < 2048; b += 256) hist[b] = 0;
    for (int i = tid; i < sort_n; i += 256) keys[i] = ~0ull;
    if (tid == 0) { s_count = 0; s_ncand = 0; s_bin = 0; s_need = 0; }
    __syncthreads();
    topk_scan_row(row, V, [&](float x, int) { atomicAdd(&hist[desc_key(x) >> 21], 1u); });
    __syncthreads();
    // locate the bin of the k-th best: 8 consecutive bins per thread + block scan
    uint32_t local[8], sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { local[q] = hist[tid * 8 + q]; sum += local[q]; }
    scan_tmp[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = (tid >= off) ? scan_tmp[tid - off] : 0;
        __syncthreads();
        scan_tmp[tid] += v;
        __syncthreads();
    }
    {
        uint32_t before = scan_tmp[tid] - sum;          // elements in bins of earlier threads
        if (before < (uint32_t)k && before + sum >= (uint32_t)k) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (before < (uint32_t)k && before + local[q] >= (uint32_t)k) {
                    s_bin = tid * 8 + q;
                    s_need = (uint32_t)k - before;       // how many to take from this bin
                }
                before += local[q];
            }
        }
    }
    __syncthreads();
    const uint32_t tbin = s_bin, need = s_need;
    const bool fits = hist[tbin] <= (uint32_t)kTopKCand;    // workgroup-uniform

    if (fits) {
        // ---- pass 2: winners + candidates ---------------------------------------
        topk_scan_row(row, V, [&](float x, int e) {
            const uint32_t key = desc_key(x);
            const uint32_t bin = key >> 21;
            if (bin < tbin) {
                keys[atomicAdd(&s_count, 1u)] = ((unsigned long long)key << 32) | (uint32_t)e;
            } else if (bin == tbin) {
                cand[atomicAdd(&s_ncand, 1u)] = ((unsigned long long)key << 32) | (uint32_t)e;
            }
        });
        __syncthreads();
        const int nc = (int)s_ncand;
        int cn = 2;
        while (cn < nc) cn <<= 1;
        for (int i = nc + tid; i < cn; i += 256) cand[i] = ~0ull;
        __syncthreads();
        for (int size = 2; size <= cn; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tid; i < cn / 2; i += 256) {
                    const int lo = 2 * i - (i & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned long long a = cand[lo], b = cand[hi];
                    if ((a > b) == up) { cand[lo] = b; cand[hi] = a; }
                }
                __syncthreads();
            }
        }
        const uint32_t base = s_count;
        for (int i = tid; i < (int)need; i += 256) keys[base + i] = cand[i];
        __syncthreads();
    } else {
        // ---- fallback: 4 x 8-bit radix select straight from global memory --------
        if (tid == 0) { s_prefix = 0; s_krem = (uint32_t)k; }
        __syncthreads();
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const int shift_hi = 8 * (pass + 1);
            topk_scan_row(row, V, [&](float x, int) {
                const uint32_t key = desc_key(x);
                const bool match = (pass == 3) || ((key >> shift_hi) == prefix);
                if (match) atomicAdd(&hist[(key >> (8 * pass)) & 0xffu], 1u);
            });
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
        __syncthreads();
        topk_scan_row(row, V, [&](float x, int e) {
            const uint32_t key = desc_key(x);
            if (key < thr) keys[atomicAdd(&s_count, 1u)] = ((unsigned long long)key << 32) | (uint32_t)e;
        });
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
    }
    // ---- final: bitonic sort of the k winners (score desc, index asc) -------------
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
    if (thr_out) {
        if (tid == 0) thr_out[blockIdx.x] = key_to_float((uint32_t)(keys[k - 1] >> 32));
        return;
    }
    for (int i = tid; i < k; i += 256) {
        const unsigned long long kv = keys[i];
        idx_out[(size_t)blockIdx.x * k + i] = (int32_t)(uint32_t)kv;
        const float cosv = key_to_float((uint32_t)(kv >> 32));
        val_out[(size_t)blockIdx.x * k + i] = (cosv + 1.0f) / 2.0f;
    }
}

// thr_out[row] = the k-th largest value of the row (raw cosine), nothing else: 4 x 8-bit
// radix select straight from (cache-resident) global memory, 1 KB of LDS -- the sampled
// threshold of the fused path does not need topk_rows' candidate lists and sorts.
__global__ __launch_bounds__(256) void kth_largest_rows(const float* __restrict__ S, int V, int k,
                                                        float* __restrict__ thr_out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t scan_tmp[256];
    __shared__ uint32_t s_prefix, s_krem;
    const int tid = threadIdx.x;
    const float* row = S + (size_t)blockIdx.x * V;
    if (tid == 0) { s_prefix = 0; s_krem = (uint32_t)k; }
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const int shift_hi = 8 * (pass + 1);
        topk_scan_row(row, V, [&](float x, int) {
            const uint32_t key = desc_key(x);
            const bool match = (pass == 3) || ((key >> shift_hi) == prefix);
            if (match) atomicAdd(&hist[(key >> (8 * pass)) & 0xffu], 1u);
        });
        __syncthreads();
        // inclusive scan of the 256 bins; the bin whose cumulative count first reaches krem
        const uint32_t mine = hist[tid];
        scan_tmp[tid] = mine;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t v = (tid >= off) ? scan_tmp[tid - off] : 0;
            __syncthreads();
            scan_tmp[tid] += v;
            __syncthreads();
        }
        const uint32_t krem = s_krem;
        const uint32_t incl = scan_tmp[tid], excl = incl - mine;
        __syncthreads();
        if (excl < krem && incl >= krem) {        // exactly one thread
            s_krem = krem - excl;
            s_prefix = (prefix << 8) | (uint32_t)tid;
        }
        __syncthreads();
    }
    if (tid == 0) thr_out[blockIdx.x] = key_to_float(s_prefix);
}

// thr_out[row] ~ the k-th largest value of the row, k <= 64: the k-th largest of the 256
// per-thread maxima.  The top k of a few thousand values almost surely sit in k different
// threads' strided subsets, so this lands within a few ranks of the exact answer -- good enough for
// the fused path, whose threshold only steers the candidate COUNT (the count checks and the
// exact fallback keep the result exact whatever the threshold).  One read of the row, one
// 256-key bitonic sort; the exact radix select above pays four passes of LDS atomics that pile
// onto a handful of bins (cosines share their leading key bits).
__global__ __launch_bounds__(256) void approx_kth_rows(const float* __restrict__ S, int V, int k,
                                                       float* __restrict__ thr_out) {
    __shared__ uint32_t keys[256];
    const int tid = threadIdx.x;
    const float* row = S + (size_t)blockIdx.x * V;
    uint32_t best = 0xffffffffu;                       // desc_key: smaller = larger score
    topk_scan_row(row, V, [&](float x, int) { best = min(best, desc_key(x)); });
    keys[tid] = best;
    __syncthreads();
    for (int size = 2; size <= 256; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < 128) {
                const int lo = 2 * tid - (tid & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const uint32_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) thr_out[blockIdx.x] = key_to_float(keys[k - 1]);
}

// ---- fused path: GEMM with a filtering epilogue (gemm.h, EPI_FILTER) ---------------
// For large entity tables the (Q, V) score matrix is never materialised:
//   1. cosines against every kScoreStride-th entity (a 1/16 GEMM) and, per query, the
//      r-th best of that sample = a threshold that about r*16 entities will reach;
//   2. the full GEMM writes the elements that reach their row's threshold into small
//      per-(row, 64-entity group) lists (about a thousand of V per row, no atomics);
//   3. topk_from_groups gathers and sorts each row's lists (score desc, index asc) and
//      emits the k best.
// A row with an overflowed group, fewer than k or more than kCandCap candidates (a
// threshold that the sample misjudged: heavy ties, adversarial entity order) is flagged
// and recomputed by the materialising path, so the result is exact in every case.
constexpr int kScoreStride = 16;
constexpr int kCandCap = 4096;   // upper bound of the per-row candidate capacity (dynamic LDS, `ccap`)

// One workgroup per query: gather the row's per-group lists (EPI_FILTER layout) into
// LDS, sort (score desc, index asc), emit the k best.  Rows with an overflowed group,
// fewer than k candidates or more than kCandCap are flagged for the materialising path.
__global__ __launch_bounds__(256) void topk_from_groups(const unsigned long long* __restrict__ cand,
                                                        const unsigned char* __restrict__ gcnt, int ngroups,
                                                        int gcap, int k, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ val_out, int q_base,
                                                        int* __restrict__ nflag, int* __restrict__ flag_list,
                                                        int ccap) {
    // ccap (a power of two <= kCandCap) keys of dynamic LDS: sized by the host for the expected
    // candidate count, so that 8 workgroups fit a CU instead of the 4 a 32 KB array allows
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    __shared__ unsigned scan[256];
    __shared__ unsigned s_bad;
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned char* gc = gcnt + (size_t)q * ngroups;
    // contiguous chunk of groups per thread -> exclusive offsets by a block scan
    const int per = (ngroups + 255) / 256;
    const int g0 = tid * per, g1 = min(ngroups, g0 + per);
    unsigned mine = 0;
    bool bad = false;
    for (int g = g0; g < g1; ++g) {
        const unsigned c = gc[g];
        bad |= c > (unsigned)gcap;
        mine += c;
    }
    if (tid == 0) s_bad = 0;
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
    for (int i = tid; i < k; i += 256) {
        const unsigned long long kv = keys[i];
        idx_out[(size_t)q * k + i] = (int32_t)(uint32_t)kv;
        val_out[(size_t)q * k + i] = (key_to_float((uint32_t)(kv >> 32)) + 1.0f) / 2.0f;
    }
}

// dst[i,:] = src[list[i],:]
__global__ void gather_rows_f32(const float* __restrict__ src, const int* __restrict__ list, int rows,
                                int d, float* __restrict__ dst) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < (size_t)rows * d;
         t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / d;
        dst[t] = src[(size_t)list[r] * d + (t - r * d)];
    }
}
// (idx, val)[list[i],:] = (idx_c, val_c)[i,:]
__global__ void scatter_topk_rows(const int32_t* __restrict__ idx_c, const float* __restrict__ val_c,
                                  const int* __restrict__ list, int rows, int k,
                                  int32_t* __restrict__ idx, float* __restrict__ val) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < (size_t)rows * k;
         t += (size_t)gridDim.x * blockDim.x) {
        const size_t r = t / k;
        const size_t o = (size_t)list[r] * k + (t - r * k);
        idx[o] = idx_c[t];
        val[o] = val_c[t];
    }
}

}  // namespace sert
