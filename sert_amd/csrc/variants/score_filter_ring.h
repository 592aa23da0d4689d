// Variants build only (SERT_SCORE_RING=1): the bf16 filter GEMM of the scorer, persistent along the entity axis.
// Measured SLOWER than score_filter_bf16 (kernels_score_bf16.h): 1.59 ms per 10 000 queries against 1.29 (a first form with
// register-staged entity tiles: 1.74).  The filter is bound by its compare / ballot / list-store epilogue -- more than half of
// the (row pair, 64-column group) steps of a tile hold a candidate and take the divergent path with its scattered 4- and
// 1-byte stores -- and this form runs two waves per SIMD where the tiled kernel runs six.  Exact (same epilogue function,
// same lists); kept for the LDS-DMA ring it demonstrates.
#pragma once
#include "../kernels_score_bf16.h"

namespace sert {

// ---- the same filter, persistent along the entity axis (kp = 128) -----------------------------------------------------
// score_filter_bf16 loads a 128-query tile AND a 128-entity tile for 16 MFMAs per wave, through a single LDS buffer with
// two barriers per 64 k: 4 GB of L2 -> LDS traffic per call and a matrix pipe at 13 % of its peak.  Here a workgroup
// keeps its 128 queries' fragments in REGISTERS (lane (row l & 31, half l >> 5): k = 16 s + 8 half .. + 7 of its row, all
// eight steps: 32 registers) and walks a contiguous range of entity tiles; only the entity tile travels, by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of three 32 kB slots, TWO tiles ahead:
// the tile for iteration t + 2 is issued at the top of iteration t, and at its end a wave waits -- counted, vmcnt(4): its own
// four DMA instructions of tile t + 2 may stay in flight -- for its share of tile t + 1 before the one barrier of the
// iteration.  (Loads return in order, so "at most four outstanding" means every older load has landed, whatever the
// epilogue's stores do to the count.)  The DMA writes LDS lane-linearly (1 kB per wave instruction = four rows of 256
// bytes): the 16-byte pieces of a row are XOR-swizzled with the row on the SOURCE side so that the ds_read_b128 fragment
// reads are conflict-free.  Same lists, same order, same results as score_filter_bf16 (same epilogue function).
__device__ __forceinline__ void sert_glds16(const void* gsrc, unsigned lds_byte_addr) {
    // (M0 = the wave-uniform LDS destination; written in the statement that uses it, restored behind it)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

__global__ __launch_bounds__(512) void score_filter_bf16_ring(const ScoreBf16Args g, int tiles_per_wg) {
    constexpr int KS = 8, SLOT = SB_T * 256;
    __shared__ __attribute__((aligned(1024))) unsigned char Bs[3 * SLOT];
    __shared__ float thr_s[SB_T];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;            // 4 x 2 waves
    const int li = lane & 31, lh = lane >> 5;
    const int tm = blockIdx.x % g.tiles_m, part = blockIdx.x / g.tiles_m;
    const int tn0 = part * tiles_per_wg, tn1 = min(g.tiles_n, tn0 + tiles_per_wg);
    if (tn0 >= tn1) return;
    const int T = tn1 - tn0;
    const int m0 = tm * SB_T;
    if (tid < SB_T) thr_s[tid] = m0 + tid < g.M ? g.thr[m0 + tid] : INFINITY;

    // the wave's 32 queries, every k step: straight from global memory into fragment registers
    bf16x8_t a[KS];
    {
        const unsigned char* pa = (const unsigned char*)g.P16 + (size_t)min(m0 + wr * 32 + li, g.M - 1) * g.kp * 2 + lh * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const bf16x8_t*>(pa + s * 32);
    }
    // DMA of one entity tile: wave w issues instructions j = 4 w .. 4 w + 3; instruction j fills rows 4 j .. 4 j + 3
    const unsigned lds_base = (unsigned)(size_t)(Bs);     // (LDS byte address of the ring)
    auto issue = [&](int t) {
        const unsigned slot = lds_base + (unsigned)(t % 3) * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 4 * w + i, row = 4 * j + (lane >> 4), c = (lane & 15) ^ (row & 15);
            const unsigned char* src = (const unsigned char*)g.E16 + (size_t)min((tn0 + t) * SB_T + row, g.N - 1) * g.estride * 2 + c * 16;
            sert_glds16(src, slot + (unsigned)j * 1024u);
        }
    };
    issue(0);
    if (T > 1) issue(1);
    // tile 0 has landed when at most the four instructions of tile 1 are outstanding (the A fragments above are older)
    if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int r0 = wc * 64 + li, r1 = r0 + 32;
    for (int t = 0; t < T; ++t) {
        if (t + 2 < T) issue(t + 2);
        const unsigned char* Bt = Bs + (t % 3) * SLOT;
        f32x16_t acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 2 * s + lh;
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(Bt + r0 * 256 + ((c ^ (r0 & 15)) << 4));
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(Bt + r1 * 256 + ((c ^ (r1 & 15)) << 4));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b1, acc[1], 0, 0, 0);
        }
        score_filter_epilogue(acc, g, tn0 + t, m0, (tn0 + t) * SB_T, wr, wc, li, lh, thr_s);
        // this wave's share of tile t + 1 has landed (tile t + 2's four instructions may still fly); every read of slot
        // t % 3 is retired (its MFMAs were issued): behind the barrier tile t + 1 may be read and slot t % 3 refilled
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

}  // namespace sert
