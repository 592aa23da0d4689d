// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, == an fmaf chain).
//
// Used for the only dense contractions on the path:
//   vectorspace  a = h.W + b (tanh epilogue)      sert/models.py:1057-1061
//                dW = h^T.da ; dh = da.W^T        (autodiff of the above)
//   loglinear    Z = G.W + b                      sert/models.py:846-849
//                dW = G^T.dZ ; dG = dZ.W^T
//   scoring      S = Q.E^T                        bin/query.py:352-357 (batched)
//
// Workgroup = 256 threads = 4 waves (2x2); tile 128x128x32; each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  Operands are
// staged through LDS k-major ([k][m] / [k][n], leading dim 132) so that a
// wave's MFMA fragment read (32 consecutive m or n for one k) is one
// conflict-free ds_read_b32 per 32-lane half.
#pragma once
#include "common.h"

namespace sert {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 32, GLD = 132;

enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_TANH = 2 };

// Source stored [k][c], contiguous along c (the tile's M or N axis).
__device__ __forceinline__ void gemm_load_kmajor(const float* __restrict__ src, int ld, int k0,
                                                 int kend, int c0, int cend, float (*dst)[GLD],
                                                 bool vec) {
    const int t = threadIdx.x;
    const int kr = t >> 3;         // 0..31
    const int cq = (t & 7) * 4;    // 0..28
    const int k = k0 + kr;
    const float* row = src + (size_t)k * ld;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cq + j * 32;
        const int gc = c0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
            if (vec && gc + 3 < cend) {
                v = *reinterpret_cast<const float4*>(row + gc);
            } else {
                if (gc + 0 < cend) v.x = row[gc + 0];
                if (gc + 1 < cend) v.y = row[gc + 1];
                if (gc + 2 < cend) v.z = row[gc + 2];
                if (gc + 3 < cend) v.w = row[gc + 3];
            }
        }
        *reinterpret_cast<float4*>(&dst[kr][c]) = v;
    }
}

// Source stored [c][k], contiguous along k: transposed on the way into LDS.
__device__ __forceinline__ void gemm_load_cmajor(const float* __restrict__ src, int ld, int k0,
                                                 int kend, int c0, int cend, float (*dst)[GLD],
                                                 bool vec) {
    const int t = threadIdx.x;
    const int c = t >> 1;           // 0..127
    const int kh = (t & 1) * 16;    // 0 / 16
    const int gc = c0 + c;
    const float* row = src + (size_t)gc * ld;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kl = kh + j * 4;
        const int k = k0 + kl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gc < cend) {
            if (vec && k + 3 < kend) {
                v = *reinterpret_cast<const float4*>(row + k);
            } else {
                if (k + 0 < kend) v.x = row[k + 0];
                if (k + 1 < kend) v.y = row[k + 1];
                if (k + 2 < kend) v.z = row[k + 2];
                if (k + 3 < kend) v.w = row[k + 3];
            }
        }
        dst[kl + 0][c] = v.x;
        dst[kl + 1][c] = v.y;
        dst[kl + 2][c] = v.z;
        dst[kl + 3][c] = v.w;
    }
}

// C[z] (M,N) = epi( op(A) (M,K) . op(B) (K,N) ) over K range of split z.
//   TA=false: A row-major (M,K)      TA=true: A stored (K,M)  -> computes A^T.B
//   TB=false: B row-major (K,N)      TB=true: B stored (N,K)  -> computes A.B^T
// gridDim = (ceil(N/128), ceil(M/128), splits); split z covers k in
// [z*kper, min(K,(z+1)*kper)) and writes to C + z*c_split_stride.
// CSB: additionally emit the column sums of op(B) over this split's k range
// (row tile 0 only) at C[z] + M*N .. +N  -- used for db = sum_i da_i, which
// rides along with the dW = h^T.da GEMM for free (the da tile is in LDS anyway).
template <bool TA, bool TB, int EPI, bool CSB = false>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const float* __restrict__ A,
                                                     const float* __restrict__ B,
                                                     float* __restrict__ C,
                                                     const float* __restrict__ bias, int M, int N,
                                                     int K, int lda, int ldb, int ldc, int kper,
                                                     size_t c_split_stride, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[GK][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[GK][GLD];
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int kbeg = blockIdx.z * kper;
    const int kend = min(K, kbeg + kper);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int li = lane & 31, lh = lane >> 5;

    float csum = 0.f;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        if (TA) gemm_load_kmajor(A, lda, k0, kend, m0, M, As, vecA);
        else    gemm_load_cmajor(A, lda, k0, kend, m0, M, As, vecA);
        if (TB) gemm_load_cmajor(B, ldb, k0, kend, n0, N, Bs, vecB);
        else    gemm_load_kmajor(B, ldb, k0, kend, n0, N, Bs, vecB);
        __syncthreads();
        if (CSB && blockIdx.y == 0) {
            const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
            float cs = 0.f;
#pragma unroll
            for (int kk = 0; kk < GK / 2; ++kk) cs += Bs[half * (GK / 2) + kk][col];
            csum += cs;
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int k = kk + lh;
            const float a0 = As[k][wr * 64 + li];
            const float a1 = As[k][wr * 64 + 32 + li];
            const float b0 = Bs[k][wc * 64 + li];
            const float b1 = Bs[k][wc * 64 + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = C + (size_t)blockIdx.z * c_split_stride;
    if (CSB && blockIdx.y == 0) {
        // the k loop ended with a barrier: As is free to reuse
        const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
        if (half == 1) As[0][col] = csum;
        __syncthreads();
        if (half == 0 && n0 + col < N) Cz[(size_t)M * N + n0 + col] = csum + As[0][col];
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wc * 64 + tn * 32 + li;
            if (col >= N) continue;
            float bv = 0.f;
            if (EPI != EPI_STORE) bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) {
                    float v = acc[tm][tn][r];
                    if (EPI == EPI_BIAS) v = v + bv;
                    if (EPI == EPI_BIAS_TANH) v = tanhf(v + bv);
                    Cz[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

template <bool TA, bool TB, int EPI, bool CSB = false>
inline void launch_gemm(hipStream_t s, const float* A, const float* B, float* C, const float* bias,
                        int M, int N, int K, int lda, int ldb, int ldc, int splits = 1,
                        int kper = 0, size_t c_split_stride = 0) {
    if (splits <= 1) { splits = 1; kper = K; }
    dim3 grid(cdiv(N, GN), cdiv(M, GM), splits);
    const int vecA = (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
    const int vecB = (ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
    hipLaunchKernelGGL((gemm_f32_mfma<TA, TB, EPI, CSB>), grid, dim3(256), 0, s, A, B, C, bias, M, N, K,
                       lda, ldb, ldc, kper, c_split_stride, vecA, vecB);
}

// out[i] = sum_s part[s*stride + i] over the split-K partial slabs, in a fixed
// association (4 interleaved groups of ascending s, then a fixed 4-way add)
// => deterministic.  Elements i < n1 go to out1[i], the rest to out2[i-n1]
// (dW followed by the fused db column sums).  64 elements per workgroup.
__global__ __launch_bounds__(256) void reduce_partials(const float* __restrict__ part, int splits,
                                                       size_t stride, size_t count,
                                                       float* __restrict__ out1, size_t n1,
                                                       float* __restrict__ out2) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + l;
    float a = 0.f;
    if (i < count) {
#pragma unroll 8
        for (int s = g; s < splits; s += 4) a += part[(size_t)s * stride + i];
    }
    red[g][l] = a;
    __syncthreads();
    if (g == 0 && i < count) {
        const float v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        if (i < n1) out1[i] = v;
        else out2[i - n1] = v;
    }
}

}  // namespace sert
