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

// Fused per-row loglinear loss: softmax over entities per token, window
// log-product, renormalisation, clipped cross-entropy and the whole backward to
// dL/dZ, with the row's (n, V) logit slab held in LDS -- ONE read of Z and ONE
// write of dZ instead of the ~9 passes of ll_softmax_rows + ll_window.  Used
// when n*V floats (+V) fit the 160 KB LDS (C1 / C2 shapes); same maths, same
// citations as the two kernels above.
//   dynamic LDS: S[n*V] | J[V]
// slot (optional, with Zu): token (i, k) reads its LOG-PROBABILITIES from Zu[slot[i*n + k], :]
// -- the table computed ONCE per distinct word of the batch (logits GEMM +
// ll_logsoftmax_rows).  The backward then stops one level earlier: since P and the clip
// mask of a token depend on its WORD only,
//     sum over the occurrences (i,k) of word u of  dZ_ke = mask_ue dJ_ie - P_ue r_ik
//   = mask_ue (sum_occ dJ_ie) - P_ue (sum_occ r_ik),
// so the kernel emits dJ_i (V floats per batch row, into Z[i*V ..]) and the n scalars r_ik
// instead of n rows of dL/dZ: 1/n of the bytes.  ll_dzu_combine finishes per word.  slot == nullptr: the
// logits are read from Z itself and overwritten in place.
template <bool TRAIN, int NT>
__global__ __launch_bounds__(NT) void ll_fused_row(float* __restrict__ Z,
                                                    const float* __restrict__ Zu,
                                                    const int32_t* __restrict__ slot,
                                                    const int32_t* __restrict__ y_int,
                                                    const int64_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices,
                                                    const float* __restrict__ data,
                                                    const float* __restrict__ w,
                                                    float* __restrict__ rowloss, int n, int V,
                                                    float inv_batch, float* __restrict__ r_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = NT / 64;
    __shared__ float red[NW];
    float* S = lds;                       // (n, V) logits -> log-probabilities
    float* Jl = lds + (size_t)n * V;      // (V) window log-product -> dJ
    const int i = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* Zi = Z + (size_t)i * n * V;
    const int total = n * V;

    // 1. slab -> LDS: batches of 8 independent 16-byte loads per thread in flight
    //    (a load->ds_write chain per iteration would expose the full HBM latency)
    if (slot != nullptr && (V & 3) == 0 && n <= V) {
        // per token: the row of its word in the distinct-word table.  The n slots go to LDS
        // first (the J area is free until phase 3), then all threads stream the n rows as one
        // flat sequence of 16-byte pieces, 8 loads in flight each.
        int* s_slot = reinterpret_cast<int*>(Jl);
        if (tid < n) s_slot[tid] = slot[(size_t)i * n + tid];
        __syncthreads();
        const int V4 = V >> 2, total4 = n * V4;
        const float inv_v4 = 1.0f / (float)V4;
        for (int q0 = tid; q0 < total4; q0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * NT;
                if (q < total4) {
                    int k = (int)((float)q * inv_v4);          // q / V4, fixed up below
                    k -= (k * V4 > q);
                    k += ((k + 1) * V4 <= q);
                    v[u] = reinterpret_cast<const float4*>(Zu + (size_t)s_slot[k] * V)[q - k * V4];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * NT;
                if (q < total4) *reinterpret_cast<float4*>(S + 4 * q) = v[u];
            }
        }
    } else if (slot != nullptr) {
        for (int t = tid; t < total; t += NT) {
            const int k = t / V;
            S[t] = Zu[(size_t)slot[(size_t)i * n + k] * V + (t - k * V)];
        }
    } else if ((total & 3) == 0) {
        for (int t0 = tid * 4; t0 < total; t0 += 8 * 4 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 4 * NT;
                v[u] = (t < total) ? *reinterpret_cast<const float4*>(Zi + t) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 4 * NT;
                if (t < total) *reinterpret_cast<float4*>(S + t) = v[u];
            }
        }
    } else {
        for (int t = tid; t < total; t += NT) S[t] = Zi[t];
    }
    __syncthreads();
    // 2. per-token log-softmax (one wave per token)       models.py:841
    //    kept in the LOG domain: log P = (z - max) - log(sum exp), so the window
    //    log-product needs no per-element logf and clip(P) is a clamp of log P
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    // (with a slot table the rows of Zu already hold log P: ll_logsoftmax_rows ran once per
    //  distinct word)
    if (slot == nullptr) {
        for (int k = wv; k < n; k += NW) {
            float* zk = S + (size_t)k * V;
            float tmx = -INFINITY;
#pragma unroll 8
            for (int e = lane; e < V; e += 64) tmx = fmaxf(tmx, zk[e]);
            tmx = wave_max(tmx);
            float sm = 0.f;
            // __expf = v_exp_f32(x*log2e): rel. error <= ~|x|*6e-8, far inside the 1e-5 loss tolerance;
            // the libm expf expansion made this kernel VALU-bound (12 waves/CU, 2 exps per element)
#pragma unroll 8
            for (int e = lane; e < V; e += 64) sm += __expf(zk[e] - tmx);
            sm = wave_sum(sm);
            const float lsm = logf(sm);
#pragma unroll 8
            for (int e = lane; e < V; e += 64) zk[e] = (zk[e] - tmx) - lsm;
        }
        __syncthreads();
    }
    // 3. window log-product J_e = sum_k log clip(P_ke) and its softmax   models.py:200-210
    float mx = -INFINITY;
    for (int e = tid; e < V; e += NT) {
        float a = 0.f;
#pragma unroll 5
        for (int k = 0; k < n; ++k) a += fminf(fmaxf(S[(size_t)k * V + e], LOGLO), LOGHI);
        Jl[e] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_max_n<NW>(mx, red);
    float se = 0.f;
    for (int e = tid; e < V; e += NT) se += expf(Jl[e] - mx);
    se = block_sum_n<NW>(se, red);
    // 4. loss and s = sum_e dQ_e Q_e over the label entries  models.py:289-292
    const float wi = TRAIN ? w[i] : 1.f;
    const float g = wi * inv_batch;
    float loss = 0.f, sdq = 0.f;
    int64_t l0 = 0, l1 = 1;
    if (y_int == nullptr) { l0 = indptr[i]; l1 = indptr[i + 1]; }
    for (int64_t l = l0 + tid; l < l1; l += NT) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        const float q = expf(Jl[e] - mx) / se;
        const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= yv * logf(qc);
        if (TRAIN) {
            const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
            sdq += (inside ? -(g * yv) / qc : 0.f) * q;
        }
    }
    loss = block_sum_n<NW>(loss, red);
    if (tid == 0) rowloss[i] = wi * loss;
    if (!TRAIN) return;
    sdq = block_sum_n<NW>(sdq, red);
    // 5. dJ_e = Q_e (dQ_e - s): label entries first need the un-overwritten J
    //    -> keep their (e, Q_e dQ_e) in registers, write the dense part, then add
    float fix_val[4];
    int fix_e[4];
    int nfix = 0;
    bool overflow = false;
    for (int64_t l = l0 + tid; l < l1; l += NT) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        const float q = expf(Jl[e] - mx) / se;
        const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
        const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
        if (nfix < 4) { fix_e[nfix] = e; fix_val[nfix] = q * (inside ? -(g * yv) / qc : 0.f); ++nfix; }
        else overflow = true;
    }
    (void)overflow;   // > 1024 labels on one instance is outside this kernel's contract (host checks)
    __syncthreads();
    for (int e = tid; e < V; e += NT) Jl[e] = -(expf(Jl[e] - mx) / se) * sdq;
    __syncthreads();
    for (int f = 0; f < nfix; ++f) Jl[fix_e[f]] += fix_val[f];
    __syncthreads();
    // 6. per token: dZ_k = P_k (dP_k - <dP_k, P_k>) with dP = dJ*mask/P, i.e.
    //    dZ_ke = mask_ke dJ_e - P_ke r_k,  r_k = sum_e mask_ke dJ_e;  straight to HBM
    if (slot != nullptr) {
        // distinct-word mode: emit dJ_i and the r_k; the per-word combination follows
        float* dj_out = Z + (size_t)i * V;
        for (int e = tid; e < V; e += NT) dj_out[e] = Jl[e];
        for (int k = wv; k < n; k += NW) {
            const float* lk = S + (size_t)k * V;
            float r = 0.f;
#pragma unroll 8
            for (int e = lane; e < V; e += 64) {
                const float lp = lk[e];
                r += (lp >= LOGLO && lp <= LOGHI) ? Jl[e] : 0.f;
            }
            r = wave_sum(r);
            if (lane == 0) r_out[(size_t)i * n + k] = r;
        }
        return;
    }
    for (int k = wv; k < n; k += NW) {
        const float* lk = S + (size_t)k * V;
        float r = 0.f;
#pragma unroll 8
        for (int e = lane; e < V; e += 64) {
            const float lp = lk[e];
            r += (lp >= LOGLO && lp <= LOGHI) ? Jl[e] : 0.f;
        }
        r = wave_sum(r);
        float* out = Zi + (size_t)k * V;
#pragma unroll 8
        for (int e = lane; e < V; e += 64) {
            const float lp = lk[e];
            const float dj = (lp >= LOGLO && lp <= LOGHI) ? Jl[e] : 0.f;
            out[e] = dj - __expf(lp) * r;
        }
    }
}

// Distinct-word mode without the LDS slab: the n log-probability rows of batch row i are
// read straight from the (cache-resident) per-word table, coalesced along e -- one pass.
//   J_e = sum_k clamp(logp_ke), Q = softmax(J), loss, dJ_e = Q_e (dQ_e - s)     as above
//   r_k = sum_e mask_ke dJ_e = (sum_e dJ_e) when no element of token k is clipped -- the
//         common case, detected in the same pass; only clipped tokens re-read their row.
// Emits dJ_i (into dJ_out[i*V ..]) and r_ik.  LDS: J[V] | slots[n].  TRAIN only.
template <int NT>
__global__ __launch_bounds__(NT) void ll_row_from_table(const float* __restrict__ Zu,
                                                        const int32_t* __restrict__ slot,
                                                        const int32_t* __restrict__ y_int,
                                                        const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const float* __restrict__ data,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ rowloss, int n, int V,
                                                        float inv_batch, float* __restrict__ dJ_out,
                                                        float* __restrict__ r_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = NT / 64;
    __shared__ float red[NW];
    __shared__ unsigned long long s_oob;      // bit k: token k has a clipped probability (n <= 64)
    float* Jl = lds;
    int* s_slot = reinterpret_cast<int*>(lds + V);
    const int i = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    if (tid < n) s_slot[tid] = slot[(size_t)i * n + tid];
    if (tid == 0) s_oob = 0ull;
    __syncthreads();
    // 1. J and its max; clipped-token bits
    float mx = -INFINITY;
    unsigned long long oob = 0ull;
    if ((V & 3) == 0) {
        // 16-byte loads, the rows of five window positions for TWO column chunks of the thread in
        // flight at once: the scalar loop below walks V / NT dependent trips of n four-byte loads (eight
        // round trips to the Infinity Cache per batch row at V_e = 1000 -- the kernel was bound by
        // that chain, 27 us of residency per workgroup, not by bytes).  Additions in window order per
        // element, as below: same bits.
        const int V4 = V >> 2;
#ifndef SERT_LL_ROW_GC
#define SERT_LL_ROW_GC 5    // (10: 80 staging registers, 317 us against 248 at C2 dims)
#endif
        constexpr int GC = SERT_LL_ROW_GC;
        for (int e0 = tid; e0 < V4; e0 += 2 * NT) {
            const int e1 = e0 + NT;
            const bool two = e1 < V4;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            for (int k0 = 0; k0 < n; k0 += GC) {
                float4 v0[GC], v1[GC];
#pragma unroll
                for (int q = 0; q < GC; ++q) {
                    const float4* row = reinterpret_cast<const float4*>(Zu + (size_t)s_slot[min(k0 + q, n - 1)] * V);
                    v0[q] = row[e0];
                    v1[q] = row[two ? e1 : e0];
                }
#pragma unroll
                for (int q = 0; q < GC; ++q) {
                    if (k0 + q >= n) continue;
                    const float x[8] = {v0[q].x, v0[q].y, v0[q].z, v0[q].w, v1[q].x, v1[q].y, v1[q].z, v1[q].w};
                    bool bad = false;
#pragma unroll
                    for (int c = 0; c < 8; ++c) bad = bad || (!(x[c] >= LOGLO && x[c] <= LOGHI) && (c < 4 || two));
                    if (bad) oob |= 1ull << (k0 + q);
                    a0.x += fminf(fmaxf(x[0], LOGLO), LOGHI); a0.y += fminf(fmaxf(x[1], LOGLO), LOGHI);
                    a0.z += fminf(fmaxf(x[2], LOGLO), LOGHI); a0.w += fminf(fmaxf(x[3], LOGLO), LOGHI);
                    a1.x += fminf(fmaxf(x[4], LOGLO), LOGHI); a1.y += fminf(fmaxf(x[5], LOGLO), LOGHI);
                    a1.z += fminf(fmaxf(x[6], LOGLO), LOGHI); a1.w += fminf(fmaxf(x[7], LOGLO), LOGHI);
                }
            }
            reinterpret_cast<float4*>(Jl)[e0] = a0;
            mx = fmaxf(mx, fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)));
            if (two) {
                reinterpret_cast<float4*>(Jl)[e1] = a1;
                mx = fmaxf(mx, fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
            }
        }
        __syncthreads();   // (the passes below read Jl element-strided: other threads' float4 stores)
    } else
    for (int e = tid; e < V; e += NT) {
        float a = 0.f;
        for (int k = 0; k < n; ++k) {
            const float lp = Zu[(size_t)s_slot[k] * V + e];
            if (!(lp >= LOGLO && lp <= LOGHI)) oob |= 1ull << k;
            a += fminf(fmaxf(lp, LOGLO), LOGHI);
        }
        Jl[e] = a;
        mx = fmaxf(mx, a);
    }
    if (oob) atomicOr(&s_oob, oob);           // (an OR: order-independent)
    mx = block_max_n<NW>(mx, red);
    float se = 0.f;
    for (int e = tid; e < V; e += NT) se += expf(Jl[e] - mx);
    se = block_sum_n<NW>(se, red);
    // 2. loss and s = sum_e dQ_e Q_e over the label entries  models.py:289-292
    const float wi = w[i];
    const float g = wi * inv_batch;
    float loss = 0.f, sdq = 0.f;
    int64_t l0 = 0, l1 = 1;
    if (y_int == nullptr) { l0 = indptr[i]; l1 = indptr[i + 1]; }
    float fix_val[4];
    int fix_e[4];
    int nfix = 0;
    for (int64_t l = l0 + tid; l < l1; l += NT) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        const float q = expf(Jl[e] - mx) / se;
        const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
        loss -= yv * logf(qc);
        const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
        const float qdq = q * (inside ? -(g * yv) / qc : 0.f);
        sdq += qdq;
        if (nfix < 4) { fix_e[nfix] = e; fix_val[nfix] = qdq; ++nfix; }
    }
    loss = block_sum_n<NW>(loss, red);
    if (tid == 0) rowloss[i] = wi * loss;
    sdq = block_sum_n<NW>(sdq, red);
    // 3. dJ_e = Q_e (dQ_e - s): dense part, then the label entries
    for (int e = tid; e < V; e += NT) Jl[e] = -(expf(Jl[e] - mx) / se) * sdq;
    __syncthreads();
    for (int f = 0; f < nfix; ++f) Jl[fix_e[f]] += fix_val[f];
    __syncthreads();
    float tot = 0.f;
    float* dj_out = dJ_out + (size_t)i * V;
    if ((V & 3) == 0) {
        // (same per-thread element sets as the strided loop would give? no: so the partial sums are
        //  grouped per float4 here -- `tot` only feeds r_k, to fp32 rounding)
        for (int e4 = tid; e4 < (V >> 2); e4 += NT) {
            const float4 dj = reinterpret_cast<const float4*>(Jl)[e4];
            reinterpret_cast<float4*>(dj_out)[e4] = dj;
            tot += (dj.x + dj.y) + (dj.z + dj.w);
        }
    } else
    for (int e = tid; e < V; e += NT) {
        const float dj = Jl[e];
        dj_out[e] = dj;
        tot += dj;
    }
    tot = block_sum_n<NW>(tot, red);
    // 4. r_k
    const unsigned long long any = s_oob;
    for (int k = wv; k < n; k += NW) {
        float r = tot;
        if ((any >> k) & 1ull) {               // clipped token: the masked sum, from its row
            const float* lk = Zu + (size_t)s_slot[k] * V;
            r = 0.f;
            for (int e = lane; e < V; e += 64) {
                const float lp = lk[e];
                r += (lp >= LOGLO && lp <= LOGHI) ? Jl[e] : 0.f;
            }
            r = wave_sum(r);
        }
        if (lane == 0) r_out[(size_t)i * n + k] = r;
    }
}

// The same row, ONE WAVE per batch row, everything in registers: no LDS, no workgroup barrier.
// ll_row_from_table spends a workgroup and five barrier-separated block reductions on a row whose whole
// state is V_e floats; with V_e <= 64 * 4 * E4PL a lane holds its E4PL float4 chunks of J (chunk c of
// the row lives on lane c % 64), the reductions are wave reductions, and four times as many rows are in
// flight per CU.  Same arithmetic per element (window order of the additions, clamp, expf, the label
// fix-up); the cross-lane sums are grouped differently -- fp32 reassociation of loss_i, s and r.
// V % 4 == 0, n <= 64, TRAIN only.
template <int E4PL>
__global__ __launch_bounds__(256) void ll_row_wave(const float* __restrict__ Zu, const int32_t* __restrict__ slot,
                                                   const int32_t* __restrict__ y_int, const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, const float* __restrict__ data,
                                                   const float* __restrict__ w, float* __restrict__ rowloss, int B, int n,
                                                   int V, float inv_batch, float* __restrict__ dJ_out,
                                                   float* __restrict__ r_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const int V4 = V >> 2;
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    const int my_slot = slot[(size_t)i * n + min(lane, n - 1)];
    float4 J[E4PL];
#pragma unroll
    for (int j = 0; j < E4PL; ++j) J[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long oob = 0ull;
#ifndef SERT_LL_WAVE_GC
#define SERT_LL_WAVE_GC 3   // (5: 269 us, 3: 261, 10: 356 for the loss group at C2 dims -- registers against rows in flight)
#endif
    constexpr int GC = (E4PL <= 4) ? SERT_LL_WAVE_GC : 2;      // window rows in flight: GC * E4PL float4 per lane
    for (int k0 = 0; k0 < n; k0 += GC) {
        float4 v[GC][E4PL];
#pragma unroll
        for (int q = 0; q < GC; ++q) {
            const int sl = __shfl(my_slot, min(k0 + q, n - 1));
#ifndef SERT_LL_WAVE_GLOBAL_LOADS
            // the row's byte offset is wave-uniform: a scalar offset of a buffer load (descriptor of the table in
            // SGPRs, lane offset = its chunk) -- no 64-bit address pair per chunk: loss group 259 -> 250 us at C2
            // dims (the caller takes this kernel only for tables below 4 GB)
            typedef float f4v __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zu), 0, 0xffffffff, 0x00020000);
            const int soff = __builtin_amdgcn_readfirstlane(sl) * V * 4;
#pragma unroll
            for (int j = 0; j < E4PL; ++j) {
                const int c = lane + 64 * j;
                const f4v t4 = __builtin_amdgcn_raw_buffer_load_b128(rs, (c < V4 ? c : 0) * 16, soff, 0);
                v[q][j] = make_float4(t4.x, t4.y, t4.z, t4.w);
            }
#else
            const float4* row = reinterpret_cast<const float4*>(Zu + (size_t)sl * V);
#pragma unroll
            for (int j = 0; j < E4PL; ++j) {
                const int c = lane + 64 * j;
                v[q][j] = row[c < V4 ? c : 0];
            }
#endif
        }
#pragma unroll
        for (int q = 0; q < GC; ++q) {
            if (k0 + q >= n) continue;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < E4PL; ++j) {
                const bool in = lane + 64 * j < V4;
                const float x[4] = {v[q][j].x, v[q][j].y, v[q][j].z, v[q][j].w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) bad = bad || (in && !(x[cc] >= LOGLO && x[cc] <= LOGHI));
                J[j].x += fminf(fmaxf(x[0], LOGLO), LOGHI); J[j].y += fminf(fmaxf(x[1], LOGLO), LOGHI);
                J[j].z += fminf(fmaxf(x[2], LOGLO), LOGHI); J[j].w += fminf(fmaxf(x[3], LOGLO), LOGHI);
            }
            if (bad) oob |= 1ull << (k0 + q);
        }
    }
    // 1. max, sum of exponentials
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < E4PL; ++j)
        if (lane + 64 * j < V4) mx = fmaxf(mx, fmaxf(fmaxf(J[j].x, J[j].y), fmaxf(J[j].z, J[j].w)));
    mx = wave_max(mx);
    float4 E[E4PL];
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < E4PL; ++j) {
        const bool in = lane + 64 * j < V4;
        E[j].x = in ? expf(J[j].x - mx) : 0.f; E[j].y = in ? expf(J[j].y - mx) : 0.f;
        E[j].z = in ? expf(J[j].z - mx) : 0.f; E[j].w = in ? expf(J[j].w - mx) : 0.f;
        se += (E[j].x + E[j].y) + (E[j].z + E[j].w);
    }
    se = wave_sum(se);
    // 2. loss and s = sum_e dQ_e Q_e over the label entries  models.py:289-292; the entry's owner lane
    //    keeps its fix-up (dQ_e Q_e) for step 3
    const float wi = w[i];
    const float g = wi * inv_batch;
    float loss = 0.f, sdq = 0.f;
    float4 fix[E4PL];
#pragma unroll
    for (int j = 0; j < E4PL; ++j) fix[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t l0 = 0, l1 = 1;
    if (y_int == nullptr) { l0 = indptr[i]; l1 = indptr[i + 1]; }
    for (int64_t l = l0; l < l1; ++l) {
        const int e = y_int ? y_int[i] : indices[l];
        const float yv = y_int ? 1.f : data[l];
        const int c = e >> 2, comp = e & 3;
        if ((c & 63) == lane) {                  // (one lane per entry: the sums below are wave sums of one term)
            const int j = c >> 6;
            float ev = 0.f;
#pragma unroll
            for (int jj = 0; jj < E4PL; ++jj)
                if (jj == j) ev = comp == 0 ? E[jj].x : comp == 1 ? E[jj].y : comp == 2 ? E[jj].z : E[jj].w;
            const float q = ev / se;
            const float qc = fminf(fmaxf(q, SERT_CLIP_LO), SERT_CLIP_HI);
            loss -= yv * logf(qc);
            const bool inside = (q >= SERT_CLIP_LO) && (q <= SERT_CLIP_HI);
            const float qdq = q * (inside ? -(g * yv) / qc : 0.f);
            sdq += qdq;
#pragma unroll
            for (int jj = 0; jj < E4PL; ++jj)
                if (jj == j) {
                    if (comp == 0) fix[jj].x += qdq; else if (comp == 1) fix[jj].y += qdq;
                    else if (comp == 2) fix[jj].z += qdq; else fix[jj].w += qdq;
                }
        }
    }
    loss = wave_sum(loss);
    sdq = wave_sum(sdq);
    if (lane == 0) rowloss[i] = wi * loss;
    // 3. dJ_e = Q_e (dQ_e - s) = -Q_e s + [label entries]
    float tot = 0.f;
    float4* dj_out = reinterpret_cast<float4*>(dJ_out + (size_t)i * V);
#pragma unroll
    for (int j = 0; j < E4PL; ++j) {
        const int c = lane + 64 * j;
        if (c >= V4) continue;
        float4 dj;
        dj.x = -(E[j].x / se) * sdq + fix[j].x; dj.y = -(E[j].y / se) * sdq + fix[j].y;
        dj.z = -(E[j].z / se) * sdq + fix[j].z; dj.w = -(E[j].w / se) * sdq + fix[j].w;
        J[j] = dj;
        dj_out[c] = dj;
        tot += (dj.x + dj.y) + (dj.z + dj.w);
    }
    tot = wave_sum(tot);
    // 4. r_k: the plain total unless a probability of token k is clipped (then the masked sum, from its row)
    unsigned lo = (unsigned)oob, hi = (unsigned)(oob >> 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo |= __shfl_xor(lo, off); hi |= __shfl_xor(hi, off); }
    const unsigned long long any = ((unsigned long long)hi << 32) | lo;
    float my_r = tot;
    if (any) {
        for (int k = 0; k < n; ++k) {
            if (!((any >> k) & 1ull)) continue;
            const int sl = __shfl(my_slot, k);
            const float4* row = reinterpret_cast<const float4*>(Zu + (size_t)sl * V);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < E4PL; ++j) {
                const int c = lane + 64 * j;
                if (c >= V4) continue;
                const float4 lp = row[c];
                r += ((lp.x >= LOGLO && lp.x <= LOGHI) ? J[j].x : 0.f) + ((lp.y >= LOGLO && lp.y <= LOGHI) ? J[j].y : 0.f) +
                     ((lp.z >= LOGLO && lp.z <= LOGHI) ? J[j].z : 0.f) + ((lp.w >= LOGLO && lp.w <= LOGHI) ? J[j].w : 0.f);
            }
            r = wave_sum(r);
            if (lane == k) my_r = r;
        }
    }
    if (lane < n) r_out[(size_t)i * n + lane] = my_r;
}

// dZu[u, e] = mask_ue DJsum[u, e] - P_ue Rsum[u]   (in place over DJsum; logp = the word's
// log-probability row, mask = eps <= P <= 1-eps)
__global__ __launch_bounds__(256) void ll_dzu_combine(float* __restrict__ dZu, const float* __restrict__ logp,
                                                      const float* __restrict__ rsum, int64_t rows, int V) {
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    const int64_t total = rows * V;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = t / V;
        const float lp = logp[t];
        const float dj = (lp >= LOGLO && lp <= LOGHI) ? dZu[t] : 0.f;
        dZu[t] = dj - __expf(lp) * rsum[u];
    }
}

// In-place row log-softmax, log P = (z - max) - log(sum exp(z - max)), one wave per row --
// the per-token phase of ll_fused_row, hoisted to run ONCE per distinct word when the step
// works on the distinct-word logit table (same formula, same __expf).
template <int EPL>
__global__ __launch_bounds__(256) void ll_logsoftmax_rows(float* __restrict__ Z, int64_t rows, int V) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* z = Z + (size_t)r * V;
    if (EPL > 0) {
        // V <= 64*EPL: the row stays in registers -- one read, one write
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
        for (int u = 0; u < EPL; ++u) sm += __expf(x[u] - mx);     // exp(-inf) = 0 for the padding
        sm = wave_sum(sm);
        const float lsm = logf(sm);
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = lane + 64 * u;
            if (e < V) z[e] = (x[u] - mx) - lsm;
        }
        return;
    }
    float mx = -INFINITY;
#pragma unroll 4
    for (int e = lane; e < V; e += 64) mx = fmaxf(mx, z[e]);
    mx = wave_max(mx);
    float sm = 0.f;
#pragma unroll 4
    for (int e = lane; e < V; e += 64) sm += __expf(z[e] - mx);
    sm = wave_sum(sm);
    const float lsm = logf(sm);
#pragma unroll 4
    for (int e = lane; e < V; e += 64) z[e] = (z[e] - mx) - lsm;
}

// dst[ids[u], :] = src[u, :] (the word-table gradient rows of the batch's distinct words);
// touched (optional): touched[ids[u]] = 1.
__global__ __launch_bounds__(256) void ll_scatter_rows(const float* __restrict__ src,
                                                       const int32_t* __restrict__ ids, int64_t rows,
                                                       int d, float* __restrict__ dst,
                                                       unsigned char* __restrict__ touched) {
    const int64_t total = rows * d;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = t / d;
        const int c = (int)(t - u * d);
        const size_t w = (size_t)ids[u];
        dst[w * d + c] = src[t];
        if (touched && c == 0) touched[w] = 1;
    }
}

// ---- streaming path for entity vocabularies whose (n, V) slab does not fit LDS ----
// (BASELINE configs[3]: V_e = 100k.)  Same maths and citations as ll_window /
// ll_fused_row, organised as passes over Z in which every workgroup owns ONE
// kLlSeg-element segment of one row, so that B*n*ceil(V/kLlSeg) workgroups stream
// with 16-byte loads instead of B workgroups walking whole rows element-wise:
//   ll_s_tokstat   Z -> per (token row, segment) (max, sum exp)          1 read
//   ll_s_lse       -> log-sum-exp per token row
//   ll_s_window    Z, lse -> J (B, V) + per (row, segment) softmax partials 1 read
//   ll_s_rowloss   -> loss, (max J, sum, s) per batch row, label corrections
//   ll_s_dj        J -> dJ (dense part), ll_s_labfix adds the label entries
//   ll_s_tokr      Z, dJ -> r_k partials                                  1 read
//   ll_s_rsum      -> r_k
//   ll_s_dz        Z, dJ, r -> dZ in place                        1 read, 1 write
// All reductions are order-fixed (deterministic).
constexpr int kLlSeg = 4096;

// the 16 elements of (row, segment) owned by this thread: 4 x float4 when the
// rows are 16-byte aligned (V % 4 == 0), else 16 strided scalars
template <bool V4>
__device__ __forceinline__ void seg_load(const float* __restrict__ row, int V, int seg, float (&x)[16],
                                         float fill) {
    const int base = seg * kLlSeg;
    if (V4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + (u * 256 + threadIdx.x) * 4;
            float4 v = make_float4(fill, fill, fill, fill);
            if (e < V) v = *reinterpret_cast<const float4*>(row + e);
            x[4 * u] = v.x; x[4 * u + 1] = v.y; x[4 * u + 2] = v.z; x[4 * u + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + u * 256 + threadIdx.x;
            x[u] = (e < V) ? row[e] : fill;
        }
    }
}
template <bool V4>
__device__ __forceinline__ int seg_index(int seg, int u) {   // element index of x[u]
    return V4 ? seg * kLlSeg + ((u >> 2) * 256 + threadIdx.x) * 4 + (u & 3)
              : seg * kLlSeg + u * 256 + threadIdx.x;
}
template <bool V4>
__device__ __forceinline__ void seg_store(float* __restrict__ row, int V, int seg, const float (&x)[16]) {
    const int base = seg * kLlSeg;
    if (V4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + (u * 256 + threadIdx.x) * 4;
            if (e < V) *reinterpret_cast<float4*>(row + e) = make_float4(x[4 * u], x[4 * u + 1], x[4 * u + 2], x[4 * u + 3]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + u * 256 + threadIdx.x;
            if (e < V) row[e] = x[u];
        }
    }
}

template <bool V4>
__global__ __launch_bounds__(256) void ll_s_tokstat(const float* __restrict__ Z, int V, int nseg,
                                                    float2* __restrict__ stat) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - row * nseg);
    float x[16];
    seg_load<V4>(Z + (size_t)row * V, V, seg, x, -INFINITY);
    float mx = x[0];
#pragma unroll
    for (int u = 1; u < 16; ++u) mx = fmaxf(mx, x[u]);
    mx = block_max_256(mx, red);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += __expf(x[u] - mx);   // exp(-inf) = 0 for the padding
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) stat[blockIdx.x] = make_float2(mx, s);
}

// out[row] = max + log(sum): merge of the nseg (max, sum) partials of a row, one wave per row
__global__ __launch_bounds__(256) void ll_s_lse(const float2* __restrict__ stat, int64_t rows, int nseg,
                                                float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float2* st = stat + (size_t)r * nseg;
    float mx = -INFINITY;
    for (int s = lane; s < nseg; s += 64) mx = fmaxf(mx, st[s].x);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int s = lane; s < nseg; s += 64) sm += st[s].y * __expf(st[s].x - mx);
    sm = wave_sum(sm);
    if (lane == 0) lse[r] = mx + logf(sm);
}

template <bool V4>
__global__ __launch_bounds__(256) void ll_s_window(const float* __restrict__ Z, const float* __restrict__ lse,
                                                   int n, int V, int nseg, float* __restrict__ J,
                                                   float2* __restrict__ jstat,
                                                   const int32_t* __restrict__ slot) {
    __shared__ float red[4];
    const int64_t i = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - i * nseg);
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
    for (int k = 0; k < n; ++k) {
        // slot: Z and lse are the distinct-word tables, row = rank of the token's word
        const int64_t row = slot ? (int64_t)slot[i * n + k] : i * n + k;
        const float l = lse[row];
        float x[16];
        seg_load<V4>(Z + (size_t)row * V, V, seg, x, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] += fminf(fmaxf(x[u] - l, LOGLO), LOGHI);
    }
    seg_store<V4>(J + (size_t)i * V, V, seg, acc);
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) if (seg_index<V4>(seg, u) < V) mx = fmaxf(mx, acc[u]);
    mx = block_max_256(mx, red);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) if (seg_index<V4>(seg, u) < V) se += expf(acc[u] - mx);
    se = block_sum_256(se, red);
    if (threadIdx.x == 0) jstat[blockIdx.x] = make_float2(mx, se);
}

// per batch row: merge the J partials, loss, s = sum_e dQ_e Q_e, and the label
// entries' Q_e dQ_e (added to dJ by ll_s_labfix once ll_s_dj has written the dense part)
template <bool TRAIN>
__global__ __launch_bounds__(256) void ll_s_rowloss(const float* __restrict__ J, const float2* __restrict__ jstat,
                                                    const int32_t* __restrict__ y_int,
                                                    const int64_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices,
                                                    const float* __restrict__ data,
                                                    const float* __restrict__ w, float* __restrict__ rowloss,
                                                    float4* __restrict__ rowinfo, float* __restrict__ labfix,
                                                    int V, int nseg, float inv_batch) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float2* st = jstat + (size_t)i * nseg;
    const float* Ji = J + (size_t)i * V;
    float mx = -INFINITY;
    for (int s = tid; s < nseg; s += 256) mx = fmaxf(mx, st[s].x);
    mx = block_max_256(mx, red);
    float se = 0.f;
    for (int s = tid; s < nseg; s += 256) se += st[s].y * expf(st[s].x - mx);
    se = block_sum_256(se, red);
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
            const float qdq = q * (inside ? -(g * yv) / qc : 0.f);
            sdq += qdq;
            labfix[y_int ? (int64_t)i : l] = qdq;
        }
    }
    loss = block_sum_256(loss, red);
    if (tid == 0) rowloss[i] = wi * loss;
    if (!TRAIN) return;
    sdq = block_sum_256(sdq, red);
    if (tid == 0) rowinfo[i] = make_float4(mx, se, sdq, 0.f);
}

// Z[row, :] -= lse[row]  (logits -> log-probabilities, per (row, segment) workgroup)
__global__ __launch_bounds__(256) void ll_s_logp_rows(float* __restrict__ Z, const float* __restrict__ lse,
                                                      int V, int nseg) {
    const int64_t row = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - row * nseg);
    const float l = lse[row];
    float* z = Z + (size_t)row * V;
    const int e0 = seg * kLlSeg;
    for (int e = e0 + threadIdx.x; e < min(V, e0 + kLlSeg); e += 256) z[e] -= l;
}

// dJ_e = -Q_e s (the dense part of Q_e (dQ_e - s)), in place over J
template <bool V4>
__global__ __launch_bounds__(256) void ll_s_dj(float* __restrict__ J, const float4* __restrict__ rowinfo,
                                               int V, int nseg) {
    const int64_t i = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - i * nseg);
    const float4 info = rowinfo[i];
    float x[16];
    seg_load<V4>(J + (size_t)i * V, V, seg, x, 0.f);
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = -(expf(x[u] - info.x) / info.y) * info.z;
    seg_store<V4>(J + (size_t)i * V, V, seg, x);
}

__global__ __launch_bounds__(256) void ll_s_labfix(float* __restrict__ J, const int32_t* __restrict__ y_int,
                                                   const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices,
                                                   const float* __restrict__ labfix, int V) {
    const int i = blockIdx.x;
    int64_t l0 = 0, l1 = 1;
    if (y_int == nullptr) { l0 = indptr[i]; l1 = indptr[i + 1]; }
    for (int64_t l = l0 + threadIdx.x; l < l1; l += 256) {
        const int e = y_int ? y_int[i] : indices[l];
        J[(size_t)i * V + e] += labfix[y_int ? (int64_t)i : l];
    }
}

// r_k partials: sum over the segment of mask_ke dJ_e, mask = eps <= P_ke <= 1-eps
template <bool V4>
__global__ __launch_bounds__(256) void ll_s_tokr(const float* __restrict__ Z, const float* __restrict__ lse,
                                                 const float* __restrict__ dJ, int n, int V, int nseg,
                                                 float* __restrict__ rpart,
                                                 const int32_t* __restrict__ slot) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - row * nseg);
    const int64_t i = row / n;
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    const int64_t srow = slot ? (int64_t)slot[row] : row;   // distinct-word tables
    const float l = lse[srow];
    float x[16], dj[16];
    seg_load<V4>(Z + (size_t)srow * V, V, seg, x, INFINITY);   // padding: log p = +inf -> masked out
    seg_load<V4>(dJ + (size_t)i * V, V, seg, dj, 0.f);
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float lp = x[u] - l;
        r += (lp >= LOGLO && lp <= LOGHI) ? dj[u] : 0.f;
    }
    r = block_sum_256(r, red);
    if (threadIdx.x == 0) rpart[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void ll_s_rsum(const float* __restrict__ rpart, int64_t rows, int nseg,
                                                 float* __restrict__ r) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a = 0.f;
    for (int s = lane; s < nseg; s += 64) a += rpart[(size_t)row * nseg + s];
    a = wave_sum(a);
    if (lane == 0) r[row] = a;
}

// dZ_ke = mask_ke dJ_e - P_ke r_k, in place over Z
template <bool V4>
__global__ __launch_bounds__(256) void ll_s_dz(float* __restrict__ Z, const float* __restrict__ lse,
                                               const float* __restrict__ dJ, const float* __restrict__ r,
                                               int n, int V, int nseg) {
    const int64_t row = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x - row * nseg);
    const int64_t i = row / n;
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    const float l = lse[row], rk = r[row];
    float x[16], dj[16];
    seg_load<V4>(Z + (size_t)row * V, V, seg, x, 0.f);
    seg_load<V4>(dJ + (size_t)i * V, V, seg, dj, 0.f);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float lp = x[u] - l;
        x[u] = ((lp >= LOGLO && lp <= LOGHI) ? dj[u] : 0.f) - __expf(lp) * rk;
    }
    seg_store<V4>(Z + (size_t)row * V, V, seg, x);
}

}  // namespace sert
