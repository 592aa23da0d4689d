// fp32 GEMM on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 pieces,
//   x = x0 + x1 + x2   (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); 3 x 8 significant bits = the 24 of fp32),
// and a product a.b is the six bf16 products a_p b_q with p + q <= 2, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The three dropped products are below 2^-24 |a||b|: against float64 the six-term sum
// is 100 x closer than an fp32 fmaf chain (tests/test_gpu_gemm.py), so what is left is the fp32 accumulation
// the fp32 MFMA path (gemm.h) has as well.  Six bf16 MFMAs of depth 16 occupy a SIMD for 6 x 32 cycles where
// the fp32 MFMA needs 8 x 64 for the same depth: the dense contractions of the training step
// (sert/models.py:1057-1061 and their autodiff) are matrix-pipe-bound at d = 300 (gemm.h: 86 TF of 157)
// and load/latency-bound at d = 128, and this kernel takes 2.7 x less of the pipe.
//
// The split happens ONCE per element and workgroup, in registers, on the way from global memory into LDS
// (9 VALU instructions per two elements: v_cvt_pk_bf16_f32, two shifts/masks, v_pk_add_f32, ...): the kernel's
// interface stays fp32 in, fp32 out, nothing else on the path changes its layout.
//
//   C (M, N) = epi( A (M, K) . op(B) ),  A row-major fp32;  TB: B stored (N, K)   else: B stored (K, N)
//   TA (split-K form, K = the batch): C[z] (M, N) = A^T . B over k range z, A stored (K, M), B stored (K, N)
//
// Workgroup tile (32 WMB WAVES_M) x (32 WNB WAVES_N), K in steps of 16 (one MFMA depth), both operands'
// three planes in LDS as [plane][row][16 k] (32 bytes per row; the 16-byte half of a row is XOR-swizzled with
// bits of the row, which makes the ds_read_b128 fragment reads conflict-free), double buffered: one barrier
// per step.  The loads of step t + 1 are issued before the MFMAs of step t and split + stored in their shadow.
// An operand that is contiguous along k is loaded in 16-byte pieces; one that is contiguous along its
// row axis (B of A.B, both operands of A^T.B) by eight row-strided dwords per lane, lanes along the
// contiguous axis -- coalesced either way, and the LDS image is the same.
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm.h"

namespace sert {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));

constexpr int X3_KC = 16;

__device__ __forceinline__ unsigned x3_pack(float a, float b) {
    const x3_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf16x2));   // v_cvt_pk_bf16_f32 (nearest even)
}
// (a, b) -> the packed bf16 pairs of the three planes; the residuals are exact in fp32
__device__ __forceinline__ void x3_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = x3_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = x3_pack(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = x3_pack(sa, sb);
}

struct X3Args {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int tiles_m, tiles_n;
    int kper, splits;          // TA only: k range per split (multiple of 16)
    size_t c_split_stride;
};

// byte offset of (row, 16-byte half) inside one plane of an operand image
// (the half is swizzled with bits 2 and 3 of the row: the fragment reads -- ds_read_b128, 16 lanes per LDS cycle, 64 banks --
//  and the 16-byte stores of a row-contiguous operand -- eight consecutive rows per cycle, 32 banks -- are both
//  conflict-free; with bit 3 alone the stores were two-way conflicts: SQ_LDS_BANK_CONFLICT 23 % of the LDS cycles of
//  the split-K kernel)
__device__ __forceinline__ int x3_off(int row, int half) { return row * 32 + ((half ^ (((row >> 2) ^ (row >> 3)) & 1)) << 4); }

// PF: k steps of HBM operand loads in flight per workgroup.  1 (rounds 4-5): the loads of step t + 1 are issued at the top of
// step t.  2 (round 6, the 128 x 128-tile launches of the C2 step -- K = 128 is EIGHT steps, two workgroups per CU: 16 kB of A
// in flight per CU where the HBM latency x the CU's share of the bandwidth asks for ~40): the loads of step t + 2 are issued
// at the top of step t into the register set step t's operands left when they were stored during step t - 1.  Same k order,
// same term order: bit-identical results.
template <bool TA, bool TB, int EPI, bool CSB, int WAVES_M, int WAVES_N, int WMB, int WNB, bool VEC = true, int PF = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N <= 4 ? 2 : 1)) void gemm_x3(const X3Args g) {
    static_assert(PF == 1 || PF == 2, "one or two k steps of operand loads in flight");
    static_assert(PF == 1 || WNB <= 2, "the deeper prefetch is written for the 128 x 128 tile");
    constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    constexpr int TM = 32 * WMB * WAVES_M, TN = 32 * WNB * WAVES_N;
    constexpr int A_PLANE = TM * 32, B_PLANE = TN * 32;          // bytes
    constexpr int BUF = 3 * (A_PLANE + B_PLANE);
    // (+ 1 KB nobody reads: a thread without a piece stores there, so that no store -- and with it no load -- sits in a
    //  conditional block the compiler could sink the load into, next to its wait)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + 1024];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    // split-K form: the workgroups of one k range (its N tiles) sit 8 block indices apart -- on this part block b runs
    // on XCD b % 8, so they share an L2 and A's rows are fetched from HBM once; otherwise along M first
    // (neighbours share the B tile)
    int z = 0, tn, tm;
    if (TA) {
        // (one k range: the block index is the tile)
        const int b = blockIdx.x, tiles = g.tiles_m * g.tiles_n, grp = 8 * tiles, t = g.splits == 1 ? b : (b >> 3) % tiles;
        z = g.splits == 1 ? 0 : (b / grp) * 8 + (b & 7);
        tn = t / g.tiles_m;
        tm = t - tn * g.tiles_m;
        if (z >= g.splits) return;
    } else {
        // (k range z -- one unless the caller split a long K over partial slabs --, then along M first)
        const int tiles = g.tiles_m * g.tiles_n, t = blockIdx.x % tiles;
        z = blockIdx.x / tiles;
        tn = t / g.tiles_m;
        tm = t - tn * g.tiles_m;
    }
    const int m0 = tm * TM, n0 = tn * TN;
    const int kbeg = z * g.kper;
    const int kend = min(g.K, kbeg + g.kper);
    const int steps = (kend - kbeg + X3_KC - 1) / X3_KC;

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging: pieces of the operand tiles held in registers between load and split ----
    // k-contiguous operand (A of A.B, B stored (N, K)): piece = (row, quarter q of the 16 k) = one float4
    // row-contiguous operand: piece = (row, half h of the 16 k) = eight dwords, rows of the source 'ld' apart
    constexpr bool A_RC = TA, B_RC = TA || !TB;
    constexpr int A_PIECES = A_RC ? (TM * 2 + THREADS - 1) / THREADS : (TM * 4 + THREADS - 1) / THREADS;
    constexpr int B_PIECES = B_RC ? (TN * 2 + THREADS - 1) / THREADS : (TN * 4 + THREADS - 1) / THREADS;
    // Which MFMA block of a step a staged piece of the NEXT step is split + stored behind: A (loaded from HBM at the top of
    // the step) from block 1 on, B (L2-resident on this path; loaded behind block 0) behind the last blocks; everything
    // behind the last block when a step has only two.  A row-contiguous B goes through ONE set of eight registers, a
    // piece at a time: piece i is loaded behind block 2 i and stored behind block 2 i + 2.
    auto lim = [](int v, int hi) { return v < hi ? v : hi; };
    auto blk_a = [&](int i) { return WNB <= 2 ? WNB - 1 : lim(1 + i * (B_RC ? 2 : 1), WNB - 1); };
    auto blk_b = [&](int i) { return WNB <= 2 ? WNB - 1 : (B_RC ? lim(2 + 2 * i, WNB - 1) : lim(WNB - 2 + i / 2, WNB - 1)); };
    auto blk_bl = [&](int i) { return WNB <= 2 ? 0 : lim(2 * i, WNB - 2); };
    static_assert(!B_RC || B_PIECES <= 2, "a row-contiguous B is staged through one set of eight registers");
    float4 ra4[PF][A_RC ? 1 : A_PIECES];
    float ra8[PF][A_RC ? A_PIECES : 1][8];
    float4 rb4[B_RC ? 1 : B_PIECES];
    float rb8[PF][8];

    // Eight row-strided dwords, no condition anywhere: a 128-bit buffer descriptor in SGPRs whose range ends with row
    // kend - 1 of the source + one 32-bit lane offset per load; k >= kend is out of range and loads zero.  A column
    // beyond the operand's extent reads a neighbour's (finite or not) value: it only ever reaches rows / columns of C
    // that are not stored.  (Written with per-load conditions, the compiler branched around every load and waited
    // for each before the next: 35 us instead of 21 at C2.)
    auto load8 = [&](const float* src, int ld, int k0, int c, float (&r)[8]) {
        const unsigned bytes = (unsigned)min((long long)kend * ld * 4, (long long)0x7fffffff);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
        const unsigned base = ((unsigned)k0 * (unsigned)ld + (unsigned)c) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(base + (unsigned)j * (unsigned)ld * 4u), 0, 0));
    };
    unsigned char* const sink = lds + 2 * BUF + lane * 16;
    auto store8 = [&](unsigned char* img, int plane_bytes, int off, const float (&r)[8]) {
        uint4 pl[3];
        x3_split2(r[0], r[1], pl[0].x, pl[1].x, pl[2].x);
        x3_split2(r[2], r[3], pl[0].y, pl[1].y, pl[2].y);
        x3_split2(r[4], r[5], pl[0].z, pl[1].z, pl[2].z);
        x3_split2(r[6], r[7], pl[0].w, pl[1].w, pl[2].w);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(img + s * plane_bytes + off) = pl[s];
    };
    auto store4 = [&](unsigned char* img, int plane_bytes, int off, const float4 v) {
        uint2 pl[3];
        x3_split2(v.x, v.y, pl[0].x, pl[1].x, pl[2].x);
        x3_split2(v.z, v.w, pl[0].y, pl[1].y, pl[2].y);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(img + s * plane_bytes + off) = pl[s];
    };
    // One k-contiguous piece of four.  VEC: a 16-byte load from a clamped (always valid) address, zeroed at the store.
    // !VEC (a leading dimension or K that is no multiple of four -- the loglinear model over 715 experts): four dword
    // buffer loads, an element outside the tile's rows or beyond kend gets bit 31 of its offset set, which is out of the
    // descriptor's range and loads zero -- arithmetic, no condition around a load.
    auto load_piece = [&](const float* base, bool row_ok, unsigned row_off, int k) -> float4 {
        if (VEC) return tile_load16(base, (row_ok && k < kend) ? row_off + (unsigned)k : 0u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned bad = (unsigned)(!(row_ok && k + j < kend)) << 31;
            e[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(((row_off + (unsigned)(k + j)) * 4u) | bad), 0, 0));
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    };
    auto gload_a = [&](int k0, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int p = tid + THREADS * i;
            if (!A_RC) {
                const int row = p >> 2, q = p & 3;
                ra4[S][i] = load_piece(g.A, row < TM && m0 + row < g.M, (unsigned)(m0 + row) * (unsigned)g.lda, k0 + 4 * q);
            } else {
                const int row = p % TM, h = p / TM;
                load8(g.A, g.lda, k0 + 8 * h, m0 + row, ra8[S][i]);
            }
        }
    };
    auto lstore_a = [&](int buf, int k0, int i, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        unsigned char* As = lds + buf * BUF;
        const int p = tid + THREADS * i;
        if (!A_RC) {
            const int row = p >> 2, q = p & 3;
            const bool ok = m0 + row < g.M && k0 + 4 * q < kend, mine = TM * 4 % THREADS == 0 || row < TM;
            store4(mine ? As : sink, mine ? A_PLANE : 0, mine ? x3_off(row, q >> 1) + ((q & 1) << 3) : 0,
                   ok ? ra4[S][i] : make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
            const int row = p % TM, h = p / TM;
            const bool mine = TM * 2 % THREADS == 0 || h < 2;
            store8(mine ? As : sink, mine ? A_PLANE : 0, mine ? x3_off(row, h) : 0, ra8[S][i]);
        }
    };
    // k-contiguous B: every piece at once
    auto gload_b4 = [&](int k0) {
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            const int p = tid + THREADS * i, row = p >> 2, q = p & 3;
            rb4[i] = load_piece(g.B, row < TN && n0 + row < g.N, (unsigned)(n0 + row) * (unsigned)g.ldb, k0 + 4 * q);
        }
    };
    auto lstore_b4 = [&](int buf, int k0, int i) {
        unsigned char* Bs = lds + buf * BUF + 3 * A_PLANE;
        const int p = tid + THREADS * i, row = p >> 2, q = p & 3;
        const bool ok = n0 + row < g.N && k0 + 4 * q < kend, mine = TN * 4 % THREADS == 0 || row < TN;
        store4(mine ? Bs : sink, mine ? B_PLANE : 0, mine ? x3_off(row, q >> 1) + ((q & 1) << 3) : 0,
               ok ? rb4[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    };
    // row-contiguous B: piece i
    static_assert(PF == 1 || !B_RC || B_PIECES == 1, "two steps in flight: one piece of a row-contiguous B per thread and step");
    auto gload_b8 = [&](int k0, int i, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        const int p = tid + THREADS * i, row = p % TN, h = p / TN;
        load8(g.B, g.ldb, k0 + 8 * h, n0 + row, rb8[S]);
    };
    float csum = 0.f;   // CSB: this thread's share of a column sum of B (its column, its half of every 16 k)
    auto lstore_b8 = [&](int buf, int i, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        unsigned char* Bs = lds + buf * BUF + 3 * A_PLANE;
        const int p = tid + THREADS * i, row = p % TN, h = p / TN;
        const bool mine = TN * 2 % THREADS == 0 || h < 2;
        if (CSB) csum += ((rb8[S][0] + rb8[S][1]) + (rb8[S][2] + rb8[S][3])) + ((rb8[S][4] + rb8[S][5]) + (rb8[S][6] + rb8[S][7]));
        store8(mine ? Bs : sink, mine ? B_PLANE : 0, mine ? x3_off(row, h) : 0, rb8[S]);
    };

    // fragment of a 32 x 16 block: lane -> row li, k = 8 lh .. 8 lh + 7 (16 bytes)
    const int a_frag = x3_off(wm * (WMB * 32) + li, lh);
    const int b_frag = x3_off(wn * (WNB * 32) + li, lh);

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;      // (PF = 1: the one register set)
    gload_a(kbeg, S0{});
    if (PF == 2) gload_a(kbeg + X3_KC, S1{});            // (beyond kend: zeros / a clamped address, never stored)
    if (!B_RC) {
        gload_b4(kbeg);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) lstore_b4(0, kbeg, i);
    } else if (PF == 2) {
        gload_b8(kbeg, 0, S0{});
        gload_b8(kbeg + X3_KC, 0, S1{});
        lstore_b8(0, 0, S0{});
    } else {
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) { gload_b8(kbeg, i, S0{}); lstore_b8(0, i, S0{}); }
    }
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) lstore_a(0, kbeg, i, S0{});
    __syncthreads();
    // One k step.  MORE (a compile-time flag: the last step is peeled) -- with a run-time `more` the compiler has to
    // assume a staged load may still be pending from a path on which its store was skipped, and puts a vmcnt(0) in
    // front of every batch of loads: the full HBM latency of A, once per step.
    // LOAD: there is a step t + PF, whose operands are requested at the top of this one into register set SLOT (= t % PF, a
    // compile-time constant: the loop below is unrolled by PF); the operands of step t + 1 are split and stored from set
    // (SLOT + 1) % PF.  PF = 1: LOAD = MORE, one set.
    auto step = [&](int t, auto more_tag, auto load_tag, auto slot_tag) {
        constexpr bool MORE = decltype(more_tag)::value, LOAD = decltype(load_tag)::value;
        using SL = std::integral_constant<int, decltype(slot_tag)::value>;           // loads of step t + PF
        using SS = std::integral_constant<int, (decltype(slot_tag)::value + 1) % PF>;   // stores of step t + 1
        const int k0 = kbeg + t * X3_KC;
        if (LOAD) {
            gload_a(k0 + PF * X3_KC, SL{});   // (from HBM: PF whole steps ahead of its split)
            if (PF == 2 && B_RC) gload_b8(k0 + PF * X3_KC, 0, SL{});
        }
        const unsigned char* As = lds + (t & 1) * BUF;
        const unsigned char* Bs = As + 3 * A_PLANE;
        x3_bf16x8 a[WMB][3];
#pragma unroll
        for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                a[i][s] = *reinterpret_cast<const x3_bf16x8*>(As + s * A_PLANE + a_frag + i * (32 * 32));
#pragma unroll
        for (int j = 0; j < WNB; ++j) {
            x3_bf16x8 b[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const x3_bf16x8*>(Bs + s * B_PLANE + b_frag + j * (32 * 32));
            // the six products with p + q <= 2, smallest first; consecutive MFMAs alternate accumulators
#define SERT_X3_TERM(P, Q)                                                                            \
    _Pragma("unroll") for (int i = 0; i < WMB; ++i)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][P], b[Q], acc[i][j], 0, 0, 0);
            SERT_X3_TERM(0, 2) SERT_X3_TERM(1, 1) SERT_X3_TERM(2, 0)
            SERT_X3_TERM(0, 1) SERT_X3_TERM(1, 0) SERT_X3_TERM(0, 0)
#undef SERT_X3_TERM
            // The next step's operands are split and stored in the shadow of this step's MFMAs, piece by piece over the
            // blocks: an MFMA leaves its SIMD's issue port free for ~5 instructions, and a wave issues in order -- with
            // the split gathered behind a block's last MFMA the matrix pipe idled for ~200 VALU instructions per wave
            // and step (both waves of a SIMD are in the same phase: the barrier sees to that).  A comes from HBM and
            // was loaded at the top of the step; B (L2) is loaded behind block 0.
            int valu = 0;
            if (MORE) {
                const int nb = (t + 1) & 1, kn = k0 + X3_KC;
#pragma unroll
                for (int i = 0; i < A_PIECES; ++i)
                    if (j == blk_a(i)) { lstore_a(nb, kn, i, SS{}); valu += A_RC ? 80 : 40; }
                if (!B_RC) {
                    if (j == 0) gload_b4(kn);
#pragma unroll
                    for (int i = 0; i < B_PIECES; ++i)
                        if (j == blk_b(i)) { lstore_b4(nb, kn, i); valu += 40; }
                } else if (PF == 2) {
                    if (j == blk_b(0)) { lstore_b8(nb, 0, SS{}); valu += 80; }
                } else {
#pragma unroll
                    for (int i = 0; i < B_PIECES; ++i) {
                        if (j == blk_b(i)) { lstore_b8(nb, i, SS{}); valu += 80; }
                        if (j == blk_bl(i)) gload_b8(kn, i, SS{});
                    }
                }
            }
            // (the pattern the scheduler is asked for: MFMA, a few VALU, MFMA, ...)
            const int per = (valu + 6 * WMB - 1) / (6 * WMB);
#define SERT_X3_PATTERN(N)                                                          \
    _Pragma("unroll") for (int q = 0; q < 6 * WMB; ++q) {                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
        __builtin_amdgcn_sched_group_barrier(0x002, N, 0);                         \
    }
            if (per == 0) {} else if (per <= 4) { SERT_X3_PATTERN(4) } else if (per <= 7) { SERT_X3_PATTERN(7) }
            else if (per <= 10) { SERT_X3_PATTERN(10) } else { SERT_X3_PATTERN(14) }
#undef SERT_X3_PATTERN
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MORE) __syncthreads();
    };
    // (an EMPTY k range -- kbeg >= K, a caller's (splits, kper) with more ranges than K holds: steps <= 0 -- reads the image
    //  the prologue stored, which is all zeros then: the slab receives zeros instead of whatever buffer 1 held)
    if (PF == 1) {
        for (int t = 0; t + 1 < steps; ++t) step(t, std::true_type{}, std::true_type{}, S0{});
        step(steps > 0 ? steps - 1 : 0, std::false_type{}, std::false_type{}, S0{});
    } else {
        int t = 0;
        for (; t + 3 < steps; t += 2) {
            step(t, std::true_type{}, std::true_type{}, S0{});
            step(t + 1, std::true_type{}, std::true_type{}, S1{});
        }
        const int rest = steps - t;      // t even: <= 3 steps left, the first of them on set 0
        if (rest == 3) {
            step(t, std::true_type{}, std::true_type{}, S0{});
            step(t + 1, std::true_type{}, std::false_type{}, S1{});
            step(t + 2, std::false_type{}, std::false_type{}, S0{});
        } else if (rest == 2) {
            step(t, std::true_type{}, std::false_type{}, S0{});
            step(t + 1, std::false_type{}, std::false_type{}, S1{});
        } else {
            step(steps > 0 ? t : 0, std::false_type{}, std::false_type{}, S0{});
        }
    }

    // ---- epilogue.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a store instruction
    // writes two 128-byte row segments.  (Computing the blocks transposed, so that a lane holds four consecutive
    // columns of one row and stores 16 bytes at a time, was SLOWER -- 32 rows x 32 bytes per instruction: C2 dh 20.8 ->
    // 22.6 us, C4 projection 80.7 -> 85.5 us.)
    float* Cz = g.C + (size_t)z * g.c_split_stride;
    auto emit = [&](auto full_tag) {
        constexpr bool FULL_ROWS = decltype(full_tag)::value;   // (no row test per store)
#pragma unroll
        for (int j = 0; j < WNB; ++j) {
            const int col = n0 + wn * (WNB * 32) + j * 32 + li;
            const bool cok = col < g.N;
            float bv = 0.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_TANH) {
                bv = g.bias[cok ? col : 0];
                // (the value is needed in a register HERE: left to its first use inside a conditional store block, the
                //  wait for this load became a vmcnt(0) in every block -- each store waited for the one before it)
                asm volatile("" : "+v"(bv));
            }
            if (!cok) continue;
#pragma unroll
            for (int i = 0; i < WMB; ++i) {
                const int rbase = m0 + wm * (WMB * 32) + i * 32 + 4 * lh;
                float* Cc = Cz + (size_t)rbase * g.ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_BIAS_TANH) v = fast_tanh(v);
                    if (FULL_ROWS || rbase + dr < g.M) Cc[(size_t)dr * g.ldc] = v;
                }
            }
        }
    };
    if (m0 + TM <= g.M) emit(std::true_type{});
    else emit(std::false_type{});
    if (CSB) {
        // the column sums of B over this split's k range ride behind the M x N slab (the bias gradient of the
        // projection).  Every staged piece of B passed through a thread's registers as fp32: the thread of column c and
        // k-half h added its eight values per step; the two halves meet in LDS.  Fixed order.
        static_assert(!CSB || (TA && B_PIECES == 1), "column sums: split-K form, one piece of B per thread");
        __syncthreads();
        float* cs = reinterpret_cast<float*>(lds);
        const int row = tid % TN, h = tid / TN;
        if (h == 1) cs[row] = csum;
        __syncthreads();
        if (h == 0 && tm == 0 && n0 + row < g.N) Cz[(size_t)g.M * g.ldc + n0 + row] = csum + cs[row];
    }
}

#ifdef SERT_VARIANTS   // (measured equal: csrc/variants/gemm_x3_bres.h, SERT_X3_BRES=1)
template <bool TB, int EPI, int WAVES>
__global__ void gemm_x3_bres(const X3Args g);
#endif

// the tile shape a product of this size takes: 0 = 128 x 128 tiles over M and N; else the row-tile height of the big forms
inline int x3_tile_cols(int N) { return N <= 128 ? 128 : (N <= 256 || (N > 320 && cdiv(N, 256) * 256 <= cdiv(N, 320) * 320)) ? 256 : 320; }
inline bool x3_big_size(int M, int N, int K, int splits) {
    const int tm = N <= 128 ? 128 : 256, tn = x3_tile_cols(N);
    // at least 128 row tiles (below that too few workgroups carry the launch -- 16384 x 300 x 300: 65 us against 41) -- or,
    // for K >= 256, a tile for every CU from a wide N (the loglinear logits at 100 000 entities: 2 300 x 100 000 x 300) or
    // from the k ranges of a long K cut into partial slabs (the loglinear dG over 100 000 entities)
    return M >= 128 * tm || (K >= 256 && M >= 1024 && (long long)cdiv(M, tm) * cdiv(N, tn) * splits >= 256);
}
// (aligned: 160 tiles x k ranges or more -- 8192 x 300 x 300: 22 us against 26, 4096 x 300 x 300 with 96: 20 against 18 --;
//  unaligned: 96, the fp32 kernels then run their scalar-loader variants -- 2033 x 715 x 300: 19 us against 26)
inline bool x3_mid_size(int M, int N, int K, int splits, bool vec) {
    return K >= 256 && M >= 1024 && (long long)cdiv(M, 128) * cdiv(N, 128) * splits >= (vec ? 160 : 96);
}

// Does the shape go to this kernel?  (every 16-byte piece aligned and wholly inside or outside; offsets below 2^31 bytes)
inline bool x3_shape_ok(bool ta, bool tb, const float* A, const float* B, int M, int N, int K, int lda, int ldb, int splits) {
    if (ta) {
        // A^T.B over a long K, split-K (dW of the projection: K = the batch; the loglinear dW = G^T.dZ; the full softmax's
        // dR_e = Z^T.dp) -- or over a shorter K when the output alone has tiles for every CU (the loglinear dW at 100 000
        // entities: 300 x 100 000 over the batch's ~2 300 distinct words)
        const long long tiles = (M > 128 && M <= 320) ? cdiv(N, 160) : (long long)cdiv(M, 128) * cdiv(N, 128);
        // (a long K with hardly any workgroups -- a split count that collapsed to 1 over one or two output tiles -- would
        //  run the whole contraction on a CU or two: the fp32 kernels' persistent tiling takes those)
        return !tb && M <= 4096 && N <= (1 << 20) && ((K >= 4096 && tiles * splits >= 16) || (K >= 1024 && tiles * splits >= 128)) &&
               (size_t)K * std::max(lda, ldb) < ((size_t)1 << 29);
    }
    const bool a_vec = lda % 4 == 0 && K % 4 == 0 && ((uintptr_t)A) % 16 == 0;
    const bool b_vec = !tb || (ldb % 4 == 0 && ((uintptr_t)B) % 16 == 0);
    const bool in_range = (size_t)M * lda < ((size_t)1 << 29) && (tb ? (size_t)N * ldb : (size_t)K * ldb) < ((size_t)1 << 29);
    const bool big = a_vec && b_vec && x3_big_size(M, N, K, splits);
    // ... or, in 128 x 128 tiles, a mid-size product with enough tiles x k ranges (the loglinear GEMMs of a
    // batch of 1024 over a few hundred experts; any alignment: x3_mid_size)
    return in_range && N <= (1 << 20) && (K <= 4096 || splits > 1) && (big || x3_mid_size(M, N, K, splits, a_vec && b_vec));
}

// Two k steps of operand loads in flight for the 128 x 128-tile launches: MEASURED EQUAL OR SLOWER (round 6,
// profiles/r06_experiments.txt item 2: the three C2 GEMMs alone 24.9 / 22.3 / 21.3 us against 24.1 / 22.0 / 22.0, the step
// 0.2410-0.2421 against 0.2382-0.2404 ms) -- the launches are not waiting for more bytes in flight.  The instantiations exist
// in a -DSERT_VARIANTS build only (SERT_X3_PF=2).
#ifdef SERT_VARIANTS
inline bool x3_pf2() {
    static const bool on = variant_knob("SERT_X3_PF") && atoi(variant_knob("SERT_X3_PF")) == 2;
    return on;
}
#endif

template <bool TB, int EPI>
inline void launch_gemm_x3(hipStream_t s, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                           int lda, int ldb, int ldc, int splits, int kper, size_t c_split_stride) {
    X3Args g = {};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.kper = splits > 1 ? kper : K; g.splits = std::max(1, splits); g.c_split_stride = splits > 1 ? c_split_stride : 0;
#ifdef SERT_VARIANTS
    static const int bres_waves = variant_knob("SERT_X3_BRES") ? atoi(variant_knob("SERT_X3_BRES")) : 0;   // 8 or 4 waves per workgroup
    if (N <= 128 && K <= 128 && bres_waves && (!TB || (ldb % 4 == 0))) {
        // B resident in LDS, one workgroup per CU, every wave its own 32-row blocks: 23.7 / 21.1 us against 22.4 / 20.6 at C2
        g.tiles_m = cdiv(M, 32); g.tiles_n = 1;
        if (bres_waves == 4) SERT_LAUNCH((gemm_x3_bres<TB, EPI, 4>), dim3(std::min(256, cdiv(g.tiles_m, 4))), dim3(256), 0, s, g);
        else SERT_LAUNCH((gemm_x3_bres<TB, EPI, 8>), dim3(std::min(256, cdiv(g.tiles_m, 8))), dim3(512), 0, s, g);
        return;
    }
#endif
    const bool vec = lda % 4 == 0 && K % 4 == 0 && ((uintptr_t)A) % 16 == 0 && (!TB || (ldb % 4 == 0 && ((uintptr_t)B) % 16 == 0));
    const bool big = vec && x3_big_size(M, N, K, g.splits);
    if (N <= 128 || !big) {
        g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 128);
#ifdef SERT_VARIANTS
        if (vec && x3_pf2()) SERT_LAUNCH((gemm_x3<false, TB, EPI, false, 2, 2, 2, 2, true, 2>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(256), 0, s, g);
        else
#endif
        if (vec) SERT_LAUNCH((gemm_x3<false, TB, EPI, false, 2, 2, 2, 2>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(256), 0, s, g);
        else     SERT_LAUNCH((gemm_x3<false, TB, EPI, false, 2, 2, 2, 2, false>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(256), 0, s, g);
    } else if (x3_tile_cols(N) == 256) {
        g.tiles_m = cdiv(M, 256); g.tiles_n = cdiv(N, 256);
        SERT_LAUNCH((gemm_x3<false, TB, EPI, false, 4, 2, 2, 4>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(512), 0, s, g);
    } else {
        g.tiles_m = cdiv(M, 256); g.tiles_n = cdiv(N, 320);
        SERT_LAUNCH((gemm_x3<false, TB, EPI, false, 4, 2, 2, 5>), dim3(g.tiles_m * g.tiles_n * g.splits), dim3(512), 0, s, g);
    }
}

// C[z] (M, N) (+ N column sums of B behind it when CSB) = A^T . B over k range z; kper a multiple of 16
template <bool CSB>
inline void launch_gemm_x3_ta(hipStream_t s, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                              int ldc, int splits, int kper, size_t c_split_stride) {
    X3Args g = {};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.kper = kper; g.splits = splits; g.c_split_stride = c_split_stride;
    g.tiles_m = 1;
    if (M <= 128 && N <= 128) {
        g.tiles_n = 1;
#ifdef SERT_VARIANTS
        if (x3_pf2()) SERT_LAUNCH((gemm_x3<true, false, EPI_STORE, CSB, 2, 2, 2, 2, true, 2>), dim3(splits == 1 ? 1 : 8 * cdiv(splits, 8)), dim3(256), 0, s, g);
        else
#endif
        SERT_LAUNCH((gemm_x3<true, false, EPI_STORE, CSB, 2, 2, 2, 2>), dim3(splits == 1 ? 1 : 8 * cdiv(splits, 8)), dim3(256), 0, s, g);
    } else if (M <= 320 && M > 128) {
        // 320 x 160 tiles, ten waves of 32 x 160 (80 accumulator registers: three waves fit a SIMD)
        g.tiles_n = cdiv(N, 160);
        SERT_LAUNCH((gemm_x3<true, false, EPI_STORE, CSB, 10, 1, 1, 5>), dim3(splits == 1 ? g.tiles_n : 8 * g.tiles_n * cdiv(splits, 8)), dim3(640), 0, s, g);
    } else {
        // 128 x 128 tiles; the tiles of one k range share an XCD
        g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 128);
        SERT_LAUNCH((gemm_x3<true, false, EPI_STORE, CSB, 2, 2, 2, 2>), dim3(splits == 1 ? g.tiles_m * g.tiles_n : 8 * g.tiles_m * g.tiles_n * cdiv(splits, 8)), dim3(256), 0, s, g);
    }
}

}  // namespace sert
