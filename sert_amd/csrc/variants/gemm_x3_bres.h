// B-resident form of gemm_x3.h for K <= 128, N <= 128 (variants build only, SERT_X3_BRES=8|4): measured EQUAL OR SLOWER than the
// tiled kernel at C2 -- projection 23.7 us against 22.4, dh 21.1 against 20.6 (eight waves per workgroup; four: 30.2 / 24.8);
// 131072 rows 35.6 against 39.2, 32768 rows 17.2 against 13.5.  A launch of one block per wave has nothing to overlap its
// preload of B (96 kB per workgroup), its A round trip and its stores with; see DESIGN.md section 3.
#pragma once
#include "../gemm_x3.h"

namespace sert {

// ---- B resident: K <= 128, N <= 128 (the projection and dh at d = 128) ---------------------------------------------
// The kernel above keeps the matrix pipe busy 30 % of such a launch: eight k steps, and every workgroup's prologue and
// epilogue at the same time on every CU.  Here B's three planes -- the whole matrix, 96 kB -- are split into LDS ONCE per
// workgroup; after that barrier there is none: each wave walks its own 32-row blocks of A, takes its A fragments straight
// from global memory (lane (row l & 31, half l >> 5) owns k = 16 s + 8 half .. + 7 of its row: two float4 per k step,
// split in registers), reads B's fragments with ds_read_b128, and stores its block -- loads, MFMAs and stores of different
// waves overlap, and a wave loads its next block while it computes the current one.
template <bool TB, int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_x3_bres(const X3Args g) {
    constexpr int THREADS = 64 * WAVES, PIECES = 4096 / THREADS;
    constexpr int KS = 8, NB = 4;                       // k steps of 16, column blocks of 32
    constexpr int IMG = 128 * 32;                       // one (plane, k step) image: 128 rows x 32 bytes
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * KS * IMG];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    auto split4 = [&](const float4 v, uint2 (&pl)[3]) {
        x3_split2(v.x, v.y, pl[0].x, pl[1].x, pl[2].x);
        x3_split2(v.z, v.w, pl[0].y, pl[1].y, pl[2].y);
    };
    // ---- B -> LDS, split (zero beyond N / K).  Eight pieces of four k per thread, every load issued before the first split
    // (one piece at a time, the loop was eight L2 round trips long: 15 us) ----
    {
        float4 v[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = tid + THREADS * i;
            if (TB) {   // B stored (N, K): piece = (row n, quarter q of k step ks) = one float4
                const int n = p >> 5, r = p & 31, k = 4 * r;
                const bool ok = n < g.N && k < g.K;
                v[i] = tile_load16(g.B, ok ? (unsigned)n * (unsigned)g.ldb + (unsigned)k : 0u);
                if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {    // B stored (K, N): piece = (column n, four consecutive k), lanes along n
                const int n = p & 127, r = p >> 7, k = 4 * r;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0,
                                                                                     (int)min((long long)g.K * g.ldb * 4, (long long)0x7fffffff), 0x00020000);
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = n < g.N ? ((unsigned)(k + j) * (unsigned)g.ldb + (unsigned)n) * 4u : 0x80000000u;
                    e[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
                }
                v[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = tid + THREADS * i;
            const int n = TB ? p >> 5 : p & 127, r = TB ? p & 31 : p >> 7, ks = r >> 2, q = r & 3;
            uint2 pl[3];
            split4(v[i], pl);
#pragma unroll
            for (int s = 0; s < 3; ++s)
                *reinterpret_cast<uint2*>(Bs + (s * KS + ks) * IMG + x3_off(n, q >> 1) + ((q & 1) << 3)) = pl[s];
        }
    }
    __syncthreads();

    const int nblocks = (g.M + 31) / 32, wave = blockIdx.x * WAVES + w, nwaves = gridDim.x * WAVES;
    const int ksteps = (g.K + 15) / 16;
    float4 cur[KS][2], nxt[KS][2];
    auto load_block = [&](int blk, float4 (&r)[KS][2]) {
        const int row = min(blk * 32 + li, g.M - 1);
        const unsigned base = (unsigned)row * (unsigned)g.lda + 8u * (unsigned)lh;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = 16 * s + 8 * lh + 4 * j;
                r[s][j] = tile_load16(g.A, k < g.K ? base + 16u * (unsigned)s + 4u * (unsigned)j : 0u);
            }
    };
    const int b_frag = x3_off(li, lh);
    if (wave < nblocks) load_block(wave, cur);
    for (int blk = wave; blk < nblocks; blk += nwaves) {
        const bool more = blk + nwaves < nblocks;
        if (more) load_block(blk + nwaves, nxt);
        f32x16 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < ksteps) {
                uint4 ap[3];
                {
                    const bool k0 = 16 * s + 8 * lh < g.K, k1 = 16 * s + 8 * lh + 4 < g.K;
                    uint2 lo[3], hi[3];
                    split4(k0 ? cur[s][0] : make_float4(0.f, 0.f, 0.f, 0.f), lo);
                    split4(k1 ? cur[s][1] : make_float4(0.f, 0.f, 0.f, 0.f), hi);
#pragma unroll
                    for (int q = 0; q < 3; ++q) ap[q] = make_uint4(lo[q].x, lo[q].y, hi[q].x, hi[q].y);
                }
                x3_bf16x8 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(x3_bf16x8, ap[q]);
#pragma unroll
                for (int jp = 0; jp < NB; jp += 2) {
                    x3_bf16x8 b[2][3];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            b[jj][q] = *reinterpret_cast<const x3_bf16x8*>(Bs + (q * KS + s) * IMG + b_frag + (jp + jj) * (32 * 32));
#define SERT_X3R_TERM(P, Q)                                                                                  \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                        \
        acc[jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P], b[jj][Q], acc[jp + jj], 0, 0, 0);
                    SERT_X3R_TERM(0, 2) SERT_X3R_TERM(1, 1) SERT_X3R_TERM(2, 0)
                    SERT_X3R_TERM(0, 1) SERT_X3R_TERM(1, 0) SERT_X3R_TERM(0, 0)
#undef SERT_X3R_TERM
                }
            }
        }
        // ---- this block's 32 x N results.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
        const int rbase = blk * 32 + 4 * lh;
        const bool full = blk * 32 + 32 <= g.M;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = j * 32 + li;
            const bool cok = col < g.N;
            float bv = 0.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
                bv = g.bias[cok ? col : 0];
                asm volatile("" : "+v"(bv));      // (in a register before the conditional stores: gemm_x3's epilogue)
            }
            if (!cok) continue;
            float* Cc = g.C + (size_t)rbase * g.ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float v = acc[j][r] + bv;
                if (EPI == EPI_BIAS_TANH) v = fast_tanh(v);
                if (full || rbase + dr < g.M) Cc[(size_t)dr * g.ldc] = v;
            }
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < KS; ++s) { cur[s][0] = nxt[s][0]; cur[s][1] = nxt[s][1]; }
        }
    }
}

}  // namespace sert
