// vectorspace / LSE kernels (sert/models.py:1024-1118), gfx950.
#pragma once
#include "common.h"

namespace sert {

// ---- K2+K3: embedding gather + window mean-pool ----------------------------
// h[i,:] = (sum_k R_w[X[i,k],:]) / n         sert/models.py:180 + :226
// One thread per (row, VEC-wide column chunk): consecutive lanes read
// consecutive 16-byte pieces of one embedding row (coalesced); the (B,n,d)
// gathered tensor is never materialised.
#ifndef SERT_GATHER_GC
#define SERT_GATHER_GC 10   // window positions fetched per trip (25 us vs 35 at C2)
#endif
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void vs_gather_mean(const IdT* __restrict__ X,
                                                      const float* __restrict__ Rw,
                                                      float* __restrict__ H, int B, int n, int d) {
    const int chunks = d / VEC;
    const int64_t total = (int64_t)B * chunks;
    const float fn = (float)n;
    for (int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tid < total;
         tid += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(tid / chunks);
        const int c = (int)(tid - (int64_t)row * chunks) * VEC;
        const IdT* xr = X + (size_t)row * n;
        if (VEC == 4) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            // five window positions per trip: their ids in one burst, their rows in a second -- two
            // dependent memory round trips per trip instead of two per token (a token's id, then its
            // row: 20 hops in series for a window of 10).  Positions past the window repeat the last
            // one (no branch around a load) and are not added; the additions keep the window order.
            constexpr int GC = SERT_GATHER_GC;
            for (int k0 = 0; k0 < n; k0 += GC) {
                size_t id[GC];
#pragma unroll
                for (int q = 0; q < GC; ++q) id[q] = (size_t)xr[min(k0 + q, n - 1)];
                float4 v[GC];
#pragma unroll
                for (int q = 0; q < GC; ++q) v[q] = *reinterpret_cast<const float4*>(Rw + id[q] * d + c);
#pragma unroll
                for (int q = 0; q < GC; ++q)
                    if (k0 + q < n) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
            a.x /= fn; a.y /= fn; a.z /= fn; a.w /= fn;
            *reinterpret_cast<float4*>(H + (size_t)row * d + c) = a;
        } else {
            float a = 0.f;
            for (int k = 0; k < n; ++k) a += Rw[(size_t)xr[k] * d + c];
            H[(size_t)row * d + c] = a / fn;
        }
    }
}

// ---- K5: negative sampler (Philox4x32-10) ----------------------------------
// iid uniform entity ids with replacement, target not excluded
// (sert/models.py:961-973).  Keyed by (seed, step, global row*z + j) so the
// stream does not depend on how rows are split over ranks.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// The (pair, entity) keys of a batch as the loss kernel writes them -- cand[i, j] = j ? neg[i, j - 1] : y[i] -- from the labels and the
// negatives alone: the stable sort of the entity-gradient chain (V_e > 2048) can then run BEFORE the loss kernel, beside the forward
// (round 6; sert_hip.hip: vs_backward, early_sort).
__global__ __launch_bounds__(256) void vs_build_cand(const int32_t* __restrict__ y, const int32_t* __restrict__ neg, int B, int z,
                                                     int32_t* __restrict__ cand) {
    const int c1 = z + 1, total = B * c1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int row = i / c1, j = i - row * c1;
        cand[i] = j ? neg[(size_t)row * z + (j - 1)] : y[row];
    }
}

// zero_f / zero_b (optional): the step's other prologue work rides along -- zero `nzf4`
// float4 of small gradient buffers and `nzb16` 16-byte words of row flags -- so that a
// training step needs ONE prologue launch on the main stream instead of two memsets and
// a sampler on a side stream plus a cross-queue wait (7-11 us of idle GPU).
__global__ void vs_sample_negatives(int32_t* __restrict__ neg, int64_t count, int64_t global_offset,
                                    uint32_t num_entities, uint64_t seed, uint64_t step,
                                    float4* __restrict__ zero_f = nullptr, size_t nzf4 = 0,
                                    uint4* __restrict__ zero_b = nullptr, size_t nzb16 = 0) {
    {
        const size_t t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (size_t i = t0; i < nzf4; i += stride) zero_f[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = t0; i < nzb16; i += stride) zero_b[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // one thread owns one Philox counter = 4 consecutive GLOBAL sample indices
    const int64_t q_first = global_offset >> 2;
    const int64_t q_last = (global_offset + count - 1) >> 2;
    const int64_t quads = q_last - q_first + 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = (uint64_t)(q_first + q);
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)step,
                         (uint32_t)(step >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t local = (int64_t)(ctr * 4 + j) - global_offset;
            if (local >= 0 && local < count)
                neg[local] = (int32_t)(((uint64_t)c[j] * num_entities) >> 32);
        }
    }
}

__global__ void convert_i64_to_i32(const int64_t* __restrict__ in, int32_t* __restrict__ out,
                                   int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int32_t)in[i];
}

// ---- K6: fused NCE score + loss + gradient ----------------------------------
// 16 lanes per batch row i (16 rows per 256-thread workgroup); lane l owns the
// float4 column chunks l, l+16, ... of the d_e-wide vectors, so one candidate's
// entity row is a single coalesced 16-lane x 16-byte read and the dot product
// closes with a 4-step DPP row reduction (no LDS).  With p = clip(t)
// (models.py:1065-1068):
//   u_j   = <R_e[c_j], p>                 c_0 = y_i, c_1..z = negatives (:990, :897)
//   s_j   = clip(sigmoid(u_j), eps, 1-eps)                                (:896-900)
//   loss  = -(log s_0 + sum_{j>0} log(1 - s_j))                           (:1091-1098)
// TRAIN additionally produces, with g = w_i / B (models.py:278-282):
//   du_j  = d loss/d u_j (clip gradient mask inclusive, [upstream Clip.grad])
//   coef[i,j] = du_j, cand[i,j] = c_j   -> dR_e[c_j] += du_j * p is done by the
//           order-fixed sorted reduction in kernels_egrad.h (no atomics)
//   da    = (sum_j du_j R_e[c_j]) * [|t| <= 1-eps] * (1 - t^2)
// rowloss[i] = (TRAIN ? w_i : 1) * loss.
template <int NCH, bool TRAIN>
__global__ __launch_bounds__(256) void vs_nce(const float* __restrict__ T,
                                              const float* __restrict__ Re,
                                              const int32_t* __restrict__ y,
                                              const int32_t* __restrict__ neg,
                                              const float* __restrict__ w, float* __restrict__ DA,
                                              float* __restrict__ coef, int32_t* __restrict__ cand,
                                              float* __restrict__ rowloss, int B, int z, int de,
                                              float inv_batch, float* __restrict__ wg_loss = nullptr) {
    // wg_loss (optional): wg_loss[blockIdx.x] = sum of this workgroup's sixteen (weighted) row
    // losses, added in row order -- the first level of the loss reduction rides along instead of
    // being a launch of its own on the step's critical path
    __shared__ float wg_red[16];
    const int l = threadIdx.x & 15;
    // Rows past the end (ragged last workgroup) are computed on a clamped row and never stored:
    // no early return, so that every wave reaches the barrier below exactly once (a divergent
    // return would make a half-active wave hit it on both paths).
    const int i_raw = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = i_raw < B;
    const int i = valid ? i_raw : B - 1;
    const int chunks = de >> 2;  // de % 4 == 0 on this path
    float4 t[NCH], p[NCH], dp[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int c = l + 16 * q;
        t[q] = (c < chunks) ? *reinterpret_cast<const float4*>(T + (size_t)i * de + 4 * c)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        p[q].x = fminf(fmaxf(t[q].x, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].y = fminf(fmaxf(t[q].y, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].z = fminf(fmaxf(t[q].z, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].w = fminf(fmaxf(t[q].w, -SERT_CLIP_HI), SERT_CLIP_HI);
        dp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float wi = TRAIN ? w[i] : 1.f;
    const float g = wi * inv_batch;
    float loss = 0.f;
    for (int j = 0; j <= z; ++j) {
#if defined(SERT_VARIANTS) && defined(KO_IDS)   // timing knock-out (wrong results), variants build only
        const int e = (i * 7 + j * 13) & 2047;
#else
        const int e = (j == 0) ? y[i] : neg[(size_t)i * z + (j - 1)];
#endif
        float4 er[NCH];
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = l + 16 * q;
#if defined(SERT_VARIANTS) && defined(KO_ROWS)   // timing knock-out (wrong results), variants build only
            er[q] = make_float4(1.f * e, 0.5f, 0.25f * j, 0.f);
#else
            er[q] = (c < chunks) ? *reinterpret_cast<const float4*>(Re + (size_t)e * de + 4 * c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            part += er[q].x * p[q].x + er[q].y * p[q].y + er[q].z * p[q].z + er[q].w * p[q].w;
        }
        const float u = row16_sum(part);
#if defined(SERT_VARIANTS) && defined(KO_MATH)   // timing knock-out (wrong results), variants build only
        const float sig = u * 0.01f + 0.5f;
        const float s = fminf(fmaxf(sig, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= s;
#else
        const float sig = theano_sigmoid(u);
        const float s = fminf(fmaxf(sig, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= (j == 0) ? logf(s) : logf(1.0f - s);
#endif
        if (TRAIN) {
            const bool inside = (sig >= SERT_CLIP_LO) && (sig <= SERT_CLIP_HI);
            float du = 0.f;
            if (inside) {
                const float ds = sig * (1.0f - sig);
                du = (j == 0) ? -(g / s) * ds : (g / (1.0f - s)) * ds;
            }
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                dp[q].x += du * er[q].x; dp[q].y += du * er[q].y;
                dp[q].z += du * er[q].z; dp[q].w += du * er[q].w;
            }
#if !(defined(SERT_VARIANTS) && defined(KO_COEF))   // (knock-out: variants build only)
            if (l == 0 && valid) {
                coef[(size_t)i * (z + 1) + j] = du;
                cand[(size_t)i * (z + 1) + j] = e;
            }
#endif
        }
    }
    if (TRAIN) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = l + 16 * q;
            if (c >= chunks || !valid) continue;
            float4 o;
            o.x = (t[q].x >= -SERT_CLIP_HI && t[q].x <= SERT_CLIP_HI) ? dp[q].x * (1.0f - t[q].x * t[q].x) : 0.f;
            o.y = (t[q].y >= -SERT_CLIP_HI && t[q].y <= SERT_CLIP_HI) ? dp[q].y * (1.0f - t[q].y * t[q].y) : 0.f;
            o.z = (t[q].z >= -SERT_CLIP_HI && t[q].z <= SERT_CLIP_HI) ? dp[q].z * (1.0f - t[q].z * t[q].z) : 0.f;
            o.w = (t[q].w >= -SERT_CLIP_HI && t[q].w <= SERT_CLIP_HI) ? dp[q].w * (1.0f - t[q].w * t[q].w) : 0.f;
            *reinterpret_cast<float4*>(DA + (size_t)i * de + 4 * c) = o;
        }
    }
    if (l == 0 && valid) rowloss[i] = wi * loss;
    if (wg_loss) {
        if (l == 0) wg_red[threadIdx.x >> 4] = valid ? wi * loss : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += wg_red[r];
            wg_loss[blockIdx.x] = s;
        }
    }
}

// All candidates of a row in registers (z + 1 <= MAXC <= 16): the candidate ids arrive in one load
// (lane j of the row's sixteen holds candidate j), their entity rows in one burst, and the sigmoid /
// log / clip arithmetic runs ONCE per row with candidate j on lane j instead of once per candidate
// on all sixteen lanes.  The per-candidate kernel above walks id -> row -> dot -> sigmoid z + 1 times
// in series with two waves per SIMD in flight (B = 8192 is 2048 waves), which is what its 40 us
// were; here a wave has two dependent memory round trips in all.
template <int NCH, bool TRAIN, int MAXC>
__global__ __launch_bounds__(256) void vs_nce_regs(const float* __restrict__ T,
                                                   const float* __restrict__ Re,
                                                   const int32_t* __restrict__ y,
                                                   const int32_t* __restrict__ neg,
                                                   const float* __restrict__ w, float* __restrict__ DA,
                                                   float* __restrict__ coef, int32_t* __restrict__ cand,
                                                   float* __restrict__ rowloss, int B, int z, int de,
                                                   float inv_batch, float* __restrict__ wg_loss = nullptr) {
    __shared__ float wg_red[16];
    const int l = threadIdx.x & 15;
    const int lane = threadIdx.x & 63;
    const int i_raw = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = i_raw < B;
    const int i = valid ? i_raw : B - 1;
    const int chunks = de >> 2;
    // candidate l of this row (lanes past z repeat the last one and contribute nothing)
    const int jl = min(l, z);
    const int my_e = (jl == 0) ? y[i] : neg[(size_t)i * z + (jl - 1)];
    float4 t[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int c = l + 16 * q;
        t[q] = (c < chunks) ? *reinterpret_cast<const float4*>(T + (size_t)i * de + 4 * c)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 er[MAXC][NCH];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const int e = __shfl(my_e, (lane & 48) | u);
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = l + 16 * q;
            er[u][q] = (c < chunks) ? *reinterpret_cast<const float4*>(Re + (size_t)e * de + 4 * c)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 p[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        p[q].x = fminf(fmaxf(t[q].x, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].y = fminf(fmaxf(t[q].y, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].z = fminf(fmaxf(t[q].z, -SERT_CLIP_HI), SERT_CLIP_HI);
        p[q].w = fminf(fmaxf(t[q].w, -SERT_CLIP_HI), SERT_CLIP_HI);
    }
    float my_u = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q)
            part += er[u][q].x * p[q].x + er[u][q].y * p[q].y + er[u][q].z * p[q].z + er[u][q].w * p[q].w;
        const float uu = row16_sum(part);
        my_u = (l == u) ? uu : my_u;
    }
    const float wi = TRAIN ? w[i] : 1.f;
    const float g = wi * inv_batch;
    const bool act = l <= z;
    const float sig = theano_sigmoid(my_u);
    const float s = fminf(fmaxf(sig, SERT_CLIP_LO), SERT_CLIP_HI);
    const float term = (l == 0) ? logf(s) : logf(1.0f - s);
    const float loss = -row16_sum(act ? term : 0.f);
    if (TRAIN) {
        const bool inside = (sig >= SERT_CLIP_LO) && (sig <= SERT_CLIP_HI);
        float du = 0.f;
        if (inside && act) {
            const float ds = sig * (1.0f - sig);
            du = (l == 0) ? -(g / s) * ds : (g / (1.0f - s)) * ds;
        }
        if (act && valid) {
            coef[(size_t)i * (z + 1) + l] = du;
            cand[(size_t)i * (z + 1) + l] = my_e;
        }
        float4 dp[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) dp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const float du_u = __shfl(du, (lane & 48) | u);   // 0 for candidates past z
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                dp[q].x += du_u * er[u][q].x; dp[q].y += du_u * er[u][q].y;
                dp[q].z += du_u * er[u][q].z; dp[q].w += du_u * er[u][q].w;
            }
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = l + 16 * q;
            if (c >= chunks || !valid) continue;
            float4 o;
            o.x = (t[q].x >= -SERT_CLIP_HI && t[q].x <= SERT_CLIP_HI) ? dp[q].x * (1.0f - t[q].x * t[q].x) : 0.f;
            o.y = (t[q].y >= -SERT_CLIP_HI && t[q].y <= SERT_CLIP_HI) ? dp[q].y * (1.0f - t[q].y * t[q].y) : 0.f;
            o.z = (t[q].z >= -SERT_CLIP_HI && t[q].z <= SERT_CLIP_HI) ? dp[q].z * (1.0f - t[q].z * t[q].z) : 0.f;
            o.w = (t[q].w >= -SERT_CLIP_HI && t[q].w <= SERT_CLIP_HI) ? dp[q].w * (1.0f - t[q].w * t[q].w) : 0.f;
            *reinterpret_cast<float4*>(DA + (size_t)i * de + 4 * c) = o;
        }
    }
    if (l == 0 && valid) rowloss[i] = wi * loss;
    if (wg_loss) {
        if (l == 0) wg_red[threadIdx.x >> 4] = valid ? wi * loss : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += wg_red[r];
            wg_loss[blockIdx.x] = sum;
        }
    }
}

// Generic-width fallback (d_e % 4 != 0): one wave per row, scalar columns.
template <int NPL, bool TRAIN>
__global__ __launch_bounds__(256) void vs_nce_scalar(const float* __restrict__ T,
                                                     const float* __restrict__ Re,
                                                     const int32_t* __restrict__ y,
                                                     const int32_t* __restrict__ neg,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ DA,
                                                     float* __restrict__ coef,
                                                     int32_t* __restrict__ cand,
                                                     float* __restrict__ rowloss, int B, int z,
                                                     int de, float inv_batch) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    float t[NPL], p[NPL], dp[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int c = lane + 64 * q;
        t[q] = (c < de) ? T[(size_t)i * de + c] : 0.f;
        p[q] = fminf(fmaxf(t[q], -SERT_CLIP_HI), SERT_CLIP_HI);
        dp[q] = 0.f;
    }
    const float wi = TRAIN ? w[i] : 1.f;
    const float g = wi * inv_batch;
    float loss = 0.f;
    for (int j = 0; j <= z; ++j) {
        const int e = (j == 0) ? y[i] : neg[(size_t)i * z + (j - 1)];
        float er[NPL];
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int c = lane + 64 * q;
            er[q] = (c < de) ? Re[(size_t)e * de + c] : 0.f;
            part += er[q] * p[q];
        }
        const float u = wave_sum(part);
#if defined(SERT_VARIANTS) && defined(KO_MATH)   // timing knock-out (wrong results), variants build only
        const float sig = u * 0.01f + 0.5f;
        const float s = fminf(fmaxf(sig, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= s;
#else
        const float sig = theano_sigmoid(u);
        const float s = fminf(fmaxf(sig, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= (j == 0) ? logf(s) : logf(1.0f - s);
#endif
        if (TRAIN) {
            const bool inside = (sig >= SERT_CLIP_LO) && (sig <= SERT_CLIP_HI);
            float du = 0.f;
            if (inside) {
                const float ds = sig * (1.0f - sig);
                du = (j == 0) ? -(g / s) * ds : (g / (1.0f - s)) * ds;
            }
#pragma unroll
            for (int q = 0; q < NPL; ++q) dp[q] += du * er[q];
            if (lane == 0) {
                coef[(size_t)i * (z + 1) + j] = du;
                cand[(size_t)i * (z + 1) + j] = e;
            }
        }
    }
    if (TRAIN) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int c = lane + 64 * q;
            const bool inside = (t[q] >= -SERT_CLIP_HI) && (t[q] <= SERT_CLIP_HI);
            if (c < de) DA[(size_t)i * de + c] = inside ? dp[q] * (1.0f - t[q] * t[q]) : 0.f;
        }
    }
    if (lane == 0) rowloss[i] = wi * loss;
}

// ---- full-softmax variant (SERT_KIND_VECTORSPACE_SOFTMAX; additive) -----------
// In place on the logits Z (B, V): P = softmax(Z_i);
//   loss_i = -log clip(P[y_i], eps, 1-eps)      (same clipping as models.py:289-292)
//   TRAIN: Z_i <- dL/dZ_i = g_i * [eps <= P_y <= 1-eps] * (P - onehot(y_i)),  g_i = w_i / B
// One wave per row.  EPL > 0: the row (V <= 64*EPL) stays in registers -- one read of Z,
// one write of dZ (the 3-pass form below re-reads the row from cache twice and pays two
// libm expf per element: 206 us at C2 against a 105 us traffic floor).  __expf as in
// kernels_ll.h (rel. error ~|x|*6e-8, far inside the 1e-5 loss tolerance).
template <bool TRAIN, int EPL>
__global__ __launch_bounds__(256) void fs_softmax_ce(float* __restrict__ Z,
                                                     const int32_t* __restrict__ y,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ rowloss, int B, int V,
                                                     float inv_batch) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    float* z = Z + (size_t)i * V;
    const int yi = y[i];
    const float wi = TRAIN ? w[i] : 1.f;
    if (EPL > 0) {
        float x[EPL > 0 ? EPL : 1];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = lane + 64 * u;
            x[u] = (e < V) ? z[e] : -INFINITY;
            mx = fmaxf(mx, x[u]);
        }
        mx = wave_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            x[u] = __expf(x[u] - mx);          // exp(-inf) = 0 for the padding
            sm += x[u];
        }
        sm = wave_sum(sm);
        // the label's probability: held by lane yi % 64, element yi / 64
        float pyl = 0.f;
#pragma unroll
        for (int u = 0; u < EPL; ++u) pyl = (lane + 64 * u == yi) ? x[u] : pyl;
        const float py = wave_sum(pyl) / sm;
        const float pyc = fminf(fmaxf(py, SERT_CLIP_LO), SERT_CLIP_HI);
        if (lane == 0) rowloss[i] = -wi * logf(pyc);
        if (TRAIN) {
            const bool inside = (py >= SERT_CLIP_LO) && (py <= SERT_CLIP_HI);
            const float g = inside ? wi * inv_batch : 0.f;
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int e = lane + 64 * u;
                if (e < V) z[e] = g * (x[u] / sm - (e == yi ? 1.f : 0.f));
            }
        }
        return;
    }
    float mx = -INFINITY;
    for (int e = lane; e < V; e += 64) mx = fmaxf(mx, z[e]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int e = lane; e < V; e += 64) sm += __expf(z[e] - mx);
    sm = wave_sum(sm);
    const float py = __expf(z[yi] - mx) / sm;
    const float pyc = fminf(fmaxf(py, SERT_CLIP_LO), SERT_CLIP_HI);
    if (lane == 0) rowloss[i] = -wi * logf(pyc);
    if (TRAIN) {
        const bool inside = (py >= SERT_CLIP_LO) && (py <= SERT_CLIP_HI);
        const float g = inside ? wi * inv_batch : 0.f;
        for (int e = lane; e < V; e += 64) {
            const float p = __expf(z[e] - mx) / sm;
            z[e] = g * (p - (e == yi ? 1.f : 0.f));
        }
    }
}

// da = dp * [|t| <= 1-eps] * (1 - t^2), in place on dp   (Clip.grad + tanh')
__global__ void vs_tanh_backward(float* __restrict__ DP, const float* __restrict__ T, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float t = T[i];
        const bool inside = (t >= -SERT_CLIP_HI) && (t <= SERT_CLIP_HI);
        DP[i] = inside ? DP[i] * (1.0f - t * t) : 0.f;
    }
}

// P = clip(T), elementwise (models.py:1065-1068) -- materialised only for the
// full-softmax variant, whose logits GEMM consumes it
__global__ void vs_clip(const float* __restrict__ T, float* __restrict__ P, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        P[i] = fminf(fmaxf(T[i], -SERT_CLIP_HI), SERT_CLIP_HI);
}

// out = tanh(avg.W + b) is done by the GEMM with the EPI_BIAS_TANH epilogue.

}  // namespace sert
