// loglinear kernels (sert/models.py:804-890), gfx950.
#pragma once
#include "common.h"

namespace sert {

// G[r,:] = R_w[X[r],:], r over the B*n tokens of the batch (models.py:180, :838)
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void ll_gather_rows(const IdT* __restrict__ X,
                                                      const float* __restrict__ Rw,
                                                      float* __restrict__ G, int64_t rows, int d) {
    const int chunks = d / VEC;
    const int64_t total = rows * chunks;
    for (int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tid < total;
         tid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = tid / chunks;
        const int c = (int)(tid - r * chunks) * VEC;
        const size_t id = (size_t)X[r];
        if (VEC == 4) {
            *reinterpret_cast<float4*>(G + (size_t)r * d + c) =
                *reinterpret_cast<const float4*>(Rw + id * d + c);
        } else {
            G[(size_t)r * d + c] = Rw[id * d + c];
        }
    }
}

// In-place row softmax P = softmax(Z) (models.py:841, T.nnet.softmax:
// max-subtracted [upstream]).  One wave per row.
__global__ __launch_bounds__(256) void ll_softmax_rows(float* __restrict__ Z, int64_t rows, int V) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* z = Z + (size_t)r * V;
    float mx = -INFINITY;
    for (int e = lane; e < V; e += 64) mx = fmaxf(mx, z[e]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int e = lane; e < V; e += 64) s += expf(z[e] - mx);
    s = wave_sum(s);
    for (int e = lane; e < V; e += 64) z[e] = expf(z[e] - mx) / s;
}

// Window log-product, renormalisation, clipped cross-entropy and their
// backward, one workgroup per batch row i.  P is the (n, V) slab of row i:
//   J_e  = sum_k log clip(P_ke)                  models.py:200-201
//   Q    = softmax(J)                            models.py:210
//   loss = -sum_e Y_e log clip(Q_e)              models.py:289-292
// TRAIN (g = w_i/B): overwrites the slab with dL/dZ:
//   dQ_e = -g Y_e / clip(Q_e) * [eps<=Q_e<=1-eps]
//   dJ_e = Q_e (dQ_e - sum_e' dQ_e' Q_e')
//   dP_ke = dJ_e [eps<=P_ke<=1-eps] / clip(P_ke)
//   dZ_ke = P_ke (dP_ke - sum_e dP_ke P_ke)
// Labels: y_int (one-hot, --one_hot_classes) or CSR rows (densified per batch
// by the reference, models.py:66-89).
template <bool TRAIN>
__global__ __launch_bounds__(256) void ll_window(float* __restrict__ P, float* __restrict__ J,
                                                 const int32_t* __restrict__ y_int,
                                                 const int64_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 const float* __restrict__ data,
                                                 const float* __restrict__ w,
                                                 float* __restrict__ rowloss, int n, int V,
                                                 float inv_batch) {
    __shared__ float red[4];
    const int i = blockIdx.x;
    float* Pi = P + (size_t)i * n * V;
    float* Ji = J + (size_t)i * V;
    const int tid = threadIdx.x;

    // pass A: J and its max
    float mx = -INFINITY;
    for (int e = tid; e < V; e += 256) {
        float a = 0.f;
        for (int k = 0; k < n; ++k) {
            const float pc = fminf(fmaxf(Pi[(size_t)k * V + e], SERT_CLIP_LO), SERT_CLIP_HI);
            a += logf(pc);
        }
        Ji[e] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_max_256(mx, red);
    // pass B: normaliser
    float se = 0.f;
    for (int e = tid; e < V; e += 256) se += expf(Ji[e] - mx);
    se = block_sum_256(se, red);

    // pass C: loss and s = sum_e dQ_e Q_e over the label entries
    const float wi = TRAIN ? w[i] : 1.f;
    const float g = wi * inv_batch;
    float loss = 0.f, sdq = 0.f;
    int64_t l0 = 0, l1 = 1;
    if (y_int == nullptr) { l0 = indptr[i]; l1 = indptr[i + 1]; }
    for (int64_t l = l0 + tid; l < l1; l += 256) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        const float q = expf(Ji[e] - mx) / se;
        const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= yv * logf(qc);
        if (TRAIN) {
            const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
            const float dq = inside ? -(g * yv) / qc : 0.f;
            sdq += dq * q;
        }
    }
    loss = block_sum_256(loss, red);
    if (tid == 0) rowloss[i] = wi * loss;
    if (!TRAIN) return;
    sdq = block_sum_256(sdq, red);

    // pass D: dJ_e = Q_e (dQ_e - s); first the dense part, then the label entries
    for (int e = tid; e < V; e += 256) {
        const float q = expf(Ji[e] - mx) / se;
        Ji[e] = -q * sdq;
    }
    __syncthreads();
    for (int64_t l = l0 + tid; l < l1; l += 256) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        // recover Q_e from the stored -Q_e*s is ill-conditioned; recompute from P
        float a = 0.f;
        for (int k = 0; k < n; ++k) {
            const float pc = fminf(fmaxf(Pi[(size_t)k * V + e], SERT_CLIP_LO), SERT_CLIP_HI);
            a += logf(pc);
        }
        const float q = expf(a - mx) / se;
        const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
        const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
        const float dq = inside ? -(g * yv) / qc : 0.f;
        Ji[e] += q * dq;
    }
    __syncthreads();

    // pass E: per window slot k, dZ_k = P_k * (dP_k - <dP_k, P_k>)
    for (int k = 0; k < n; ++k) {
        float* Pk = Pi + (size_t)k * V;
        float r = 0.f;
        for (int e = tid; e < V; e += 256) {
            const float p = Pk[e];
            const bool inside = (p >= SERT_CLIP_LO) && (p <= SERT_CLIP_HI);
            if (inside) r += (Ji[e] / p) * p;
        }
        r = block_sum_256(r, red);
        for (int e = tid; e < V; e += 256) {
            const float p = Pk[e];
            const bool inside = (p >= SERT_CLIP_LO) && (p <= SERT_CLIP_HI);
            const float dp = inside ? Ji[e] / p : 0.f;
            Pk[e] = p * (dp - r);
        }
    }
}

}  // namespace sert
