// dh = da.W^T  AND  dW = h^T.da (+ db = column sums of da) in ONE launch, for the projection shape
// d_w = d_e = 128 (autodiff of sert/models.py:1055-1061).
//
// Both GEMMs of the backward consume the same da rows.  As two launches of gemm.h each of them is a
// one-tile-per-workgroup problem (65536 x 128 x 128: two tiles per CU), bound by pipeline fill and drain:
// 29 + 29 us for 4.3 GFLOP whose MFMA time is 27 us.  Here a workgroup is alone on its CU and walks
// 64-row strips: W lives in LDS for the whole launch (read once: 64 KB), a strip's da rows and h rows
// are staged once and feed BOTH products --
//   dh strip (64 x 128) = da_strip . W^T           : eight waves, one 32x32 block each
//   dW      (128 x 128) += h_strip^T . da_strip     : eight waves, two 32x32 accumulators each, which
//                                                     stay in registers over all strips of the workgroup
// -- and the next strip's rows are already in flight (registers) while the 256 MFMAs of the current one
// issue from two waves per SIMD.  LDS leading dimensions of 129 make the transposed operand
// reads (lanes along rows) conflict-free, so one copy of da serves as A operand of dh and as B operand of
// dW.  Per workgroup ONE partial dW slab + column sums (the step's tail launch combines them: 256 slabs
// instead of the 512 of the split-K GEMM).  Deterministic: fixed strip order, fixed k order.
#pragma once
#include "../common.h"
#include "../gemm.h"

namespace sert {

constexpr int FB_D = 128;          // d_w = d_e
constexpr int FB_ROWS = 64;        // rows per strip
constexpr int FB_LDW = FB_D + 1;   // W and da in LDS: odd leading dimension
constexpr int FB_LDH = FB_D + 4;   // h in LDS: 16-byte aligned rows

struct BwdFusedArgs {
    const float* DA;   // (B, 128)
    const float* H;    // (B, 128)
    const float* W;    // (128, 128)  row = d_w index, col = d_e index
    float* DH;         // (B, 128)
    float* part;       // [gridDim.x][128*128 + 128]
    int B;
    size_t stride;     // 128*128 + 128
};

constexpr int FB_THREADS = 512;    // eight waves: two per SIMD (one wave per SIMD left the matrix pipe idle
                                   // 60 % of the time behind its own LDS fragment reads: 69 us against 58 for
                                   // the two gemm.h launches)

__global__ __launch_bounds__(FB_THREADS, 1) void vs_bwd_fused(const BwdFusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) float fb_lds[];
    float (*Ws)[FB_LDW] = reinterpret_cast<float (*)[FB_LDW]>(fb_lds);                               // [128][129]
    float (*DAs)[FB_LDW] = reinterpret_cast<float (*)[FB_LDW]>(fb_lds + FB_D * FB_LDW);              // [64][129]
    float (*Hs)[FB_LDH] = reinterpret_cast<float (*)[FB_LDH]>(fb_lds + (FB_D + FB_ROWS) * FB_LDW);   // [64][132]; (128 + 64) * 129 floats is a multiple of 16 bytes
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nstrips = (g.B + FB_ROWS - 1) / FB_ROWS;

    // ---- stage W once: thread -> row t / 4, 32 consecutive columns ----
    {
        const int r = t >> 2, c0 = (t & 3) * 32;
        const float4* src = reinterpret_cast<const float4*>(g.W + (size_t)r * FB_D + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = src[j];
            Ws[r][c0 + 4 * j + 0] = v.x; Ws[r][c0 + 4 * j + 1] = v.y;
            Ws[r][c0 + 4 * j + 2] = v.z; Ws[r][c0 + 4 * j + 3] = v.w;
        }
    }
    // strip rows of a thread: row t / 8, 16 consecutive columns (4 float4) of da and of h
    const int sr = t >> 3, sc = (t & 7) * 16;
    float4 rda[4], rh[4];
    auto gload = [&](int strip) {
        const int row = strip * FB_ROWS + sr;
        const bool ok = row < g.B;
        const size_t off = (size_t)(ok ? row : 0) * FB_D + sc;
        const float4* pa = reinterpret_cast<const float4*>(g.DA + off);
        const float4* ph = reinterpret_cast<const float4*>(g.H + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rda[j] = pa[j];
            rh[j] = ph[j];
            if (!ok) { rda[j] = make_float4(0.f, 0.f, 0.f, 0.f); rh[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            DAs[sr][sc + 4 * j + 0] = rda[j].x; DAs[sr][sc + 4 * j + 1] = rda[j].y;
            DAs[sr][sc + 4 * j + 2] = rda[j].z; DAs[sr][sc + 4 * j + 3] = rda[j].w;
            *reinterpret_cast<float4*>(&Hs[sr][sc + 4 * j]) = rh[j];
        }
    };
    // dh: wave w owns the 32 x 32 block (row block w / 4, column block w % 4) of the strip
    const int xm = w >> 2, xn = w & 3;
    // dW: wave w owns row block w % 4 and the column blocks 2 (w / 4), 2 (w / 4) + 1
    const int wm = w & 3, wn0 = (w >> 2) * 2;
    f32x16 accW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;
    float colsum = 0.f;             // threads 0..127: sum over the rows of da[:, t]

    int strip = blockIdx.x;
    if (strip < nstrips) gload(strip);
    __syncthreads();                // (W staged)
    for (; strip < nstrips; strip += gridDim.x) {
        lstore();
        __syncthreads();
        const int next = strip + gridDim.x;
        if (next < nstrips) gload(next);          // in flight under the MFMAs below
        // ---- dh strip = da_strip . W^T : out[m][n] = sum_k da[m][k] W[n][k] ----
        f32x16 accX;
#pragma unroll
        for (int r = 0; r < 16; ++r) accX[r] = 0.f;
#pragma unroll 16
        for (int k = 0; k < FB_D; k += 2) {
            const float b = Ws[xn * 32 + li][k + lh];
            const float a = DAs[xm * 32 + li][k + lh];
            accX = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accX, 0, 0, 0);
        }
        {
            // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const int col = xn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = strip * FB_ROWS + xm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.B) g.DH[(size_t)row * FB_D + col] = accX[r];
            }
        }
        // ---- dW += h_strip^T . da_strip : out[m][n] += sum_row h[row][m] da[row][n] ----
#pragma unroll 16
        for (int k = 0; k < FB_ROWS; k += 2) {
            const float a = Hs[k + lh][wm * 32 + li];
            const float b0 = DAs[k + lh][wn0 * 32 + li];
            const float b1 = DAs[k + lh][wn0 * 32 + 32 + li];
            accW[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, accW[0], 0, 0, 0);
            accW[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, accW[1], 0, 0, 0);
        }
        if (t < FB_D) {
            float cs = 0.f;
#pragma unroll 16
            for (int r = 0; r < FB_ROWS; ++r) cs += DAs[r][t];
            colsum += cs;
        }
        __syncthreads();            // everyone is done reading this strip's rows
    }
    // ---- this workgroup's partial slab ----
    float* out = g.part + (size_t)blockIdx.x * g.stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = (wn0 + j) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[(size_t)row * FB_D + col] = accW[j][r];
        }
    }
    if (t < FB_D) out[(size_t)FB_D * FB_D + t] = colsum;
}

inline size_t vs_bwd_fused_lds_bytes() {
    return ((size_t)(FB_D + FB_ROWS) * FB_LDW + (size_t)FB_ROWS * FB_LDH) * sizeof(float);
}

}  // namespace sert
