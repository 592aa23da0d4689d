// Window gather + mean-pool + tanh projection in ONE launch (sert/models.py:180, :226, :1055-1061), gfx950.
//
//   h[i, :] = (sum_k R_w[X[i, k], :]) / n                      -> H (the dW GEMM of the backward reads it)
//   t[i, :] = tanh(h[i, :] . W + b)                            -> T
//
// The two launches this replaces -- vs_gather_mean (cache-bound: 655 k row fetches of 512 B at C2, 24 us) and the
// bf16-pipe GEMM gemm_x3<NN, tanh> (22 us, half of it prologue and epilogue, 33.5 MB of h written by the one and read
// by the other) -- share nothing but h, and the matrix pipe idles through the first while the memory system idles through
// much of the second.  Here ONE persistent workgroup per CU (8 waves) walks tiles of 64 batch rows, software-pipelined:
//   every wave  issues the row fetches of tile i + 1 -- four (row, 16-byte piece) items per thread, all forty fetches in flight,
//               the window's ids read from an LDS copy that was fetched one tile earlier (one memory round trip per tile
//               instead of two) --;
//   waves 4-7   then MULTIPLY tile i while those fetches are in flight: gemm_x3's main loop, two 32 x 32 blocks per wave, A
//               from the tile's image and B from an image of ALL of W's planes (96 kB, split ONCE per workgroup): no global
//               load, no barrier inside a tile; six bf16 MFMAs per fp32 product in gemm_x3's term order, fp32 accumulators
//               (bit for bit the t of the unfused path where that path runs gemm_x3:
//               tests/test_gpu_parity.py::test_fused_projection_equals_the_two_launches); bias + tanh + store;
//   every wave  behind a barrier pools what arrived exactly as vs_gather_mean does (same window order, same division: the same
//               bits of h), stores h, and splits it into the three bf16 planes of gemm_x3.h (x = x0 + x1 + x2 exactly) in the
//               (single, 48 kB) A image; a second barrier hands it to the multiplying waves.
// 150 kB of LDS.  MEASURED, NOT THE DEFAULT (opt-in, SERT_PROJ_FUSED=1): 53 us at C2 against 25 + 25 for the two launches, 18.6
// against 9 + 11.5 at 8192 rows.  The gather is bound by what a CU can pull out of L2 / the Infinity Cache (6 us per 64-row tile
// at the rate the standalone gather reaches), and here nothing is in flight while a tile is pooled, split and handed over
// (~5 us per tile); the standalone kernel has other waves fetching meanwhile.  A second A image would let the next tile's fetches
// start before the pooling -- it does not fit beside W's 96 kB.  (Round 5, on the way -- profiles/r05_experiments.txt: both phases in every wave, two workgroups per CU: 62.7 us
// at C2 with W's slices loaded per k step, 51.5 with them in registers, against 25 + 24 for the two launches -- the workgroups of
// a CU run their phases in lockstep; W's fragments in 192 registers of four multiplying waves: spills; four gathering + four
// multiplying waves over two 32-row images: 60 us, one gathering wave per SIMD does not keep enough fetches in flight.)
// Shapes: d_w % 16 == 0, d_w <= 128, d_e % 4 == 0, d_e <= 128 (one column tile); anything else takes the two launches.
#pragma once
#include "../common.h"
#include "../gemm_x3.h"

namespace sert {

constexpr int PJ_TM = 64, PJ_TN = 128, PJ_THREADS = 512;
constexpr int PJ_A_PLANE = PJ_TM * 32, PJ_B_PLANE = PJ_TN * 32;    // bytes per plane and k step
constexpr int PJ_MAX_KSTEPS = 8;                                    // d_w <= 128
constexpr int PJ_MAX_WINDOW = 10;                                   // one gather trip per item
constexpr int PJ_A_IMAGE = PJ_MAX_KSTEPS * 3 * PJ_A_PLANE;          // 49152
constexpr int PJ_W_IMAGE = PJ_MAX_KSTEPS * 3 * PJ_B_PLANE;          // 98304
constexpr int PJ_IDS = PJ_TM * PJ_MAX_WINDOW;                       // ids of one tile
constexpr int PJ_LDS = PJ_W_IMAGE + PJ_A_IMAGE + 2 * PJ_IDS * 4;    // 152576

template <typename IdT>
__global__ __launch_bounds__(PJ_THREADS, 1) void vs_project_x3(const IdT* __restrict__ X, const float* __restrict__ Rw,
                                                               const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ H, float* __restrict__ T, int B, int n, int dw,
                                                               int de) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[PJ_LDS];
    unsigned char* const Wimg = lds;
    unsigned char* const Aimg = lds + PJ_W_IMAGE;
    unsigned* const ids_lds = reinterpret_cast<unsigned*>(lds + PJ_W_IMAGE + PJ_A_IMAGE);     // [2][PJ_IDS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool multiplier = wave >= 4;             // (wave-uniform)
    const int w = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int ksteps = dw >> 4, chunks = dw >> 2;
    const int ntiles = (B + PJ_TM - 1) / PJ_TM;
    // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int mine = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const float fn = (float)n;
    const int items = PJ_TM * chunks;              // (row, float4 piece) items of a tile: four per thread at d_w = 128
    constexpr int NI = 4;

    // the ids of a tile: element p = row * n + k of the tile's rows (rows past B repeat row B - 1)
    auto load_ids = [&](int tile, unsigned (&r)[2]) {
        const int cnt = PJ_TM * n;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + u * PJ_THREADS;
            const int row = min(tile * PJ_TM + min(p, cnt - 1) / n, B - 1), k = min(p, cnt - 1) % n;
            r[u] = (unsigned)X[(size_t)row * n + k];
        }
    };
    auto store_ids = [&](int buf, const unsigned (&r)[2]) {
        const int cnt = PJ_TM * n;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + u * PJ_THREADS;
            if (p < cnt) ids_lds[buf * PJ_IDS + p] = r[u];
        }
    };

    // ---- W's planes, once per workgroup: piece = (k step, column, half of the 16 k) = eight row-strided dwords, split and stored
    // as gemm_x3 stores a row-contiguous B (its lstore_b8): image [k step][plane][column: 32 B, halves swizzled] ----
    {
        unsigned first_ids[2];
        if (mine > 0) load_ids(blockIdx.x, first_ids);
        const unsigned bytes = (unsigned)dw * (unsigned)de * 4u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)bytes, 0x00020000);
        for (int p = tid; p < PJ_MAX_KSTEPS * PJ_TN * 2; p += PJ_THREADS) {
            const int col = p % PJ_TN, h = (p / PJ_TN) & 1, ks = p / (2 * PJ_TN);
            // (a column >= de, a k row >= dw: out of the descriptor's range -> zero; nothing of such a column is stored either)
            const unsigned bad = (unsigned)(col >= de) << 31;
            const unsigned base = (((unsigned)(ks * X3_KC + 8 * h) * (unsigned)de + (unsigned)col) * 4u) | bad;
            float r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                r[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(base + (unsigned)q * (unsigned)de * 4u), 0, 0));
            uint4 pl[3];
            x3_split2(r[0], r[1], pl[0].x, pl[1].x, pl[2].x);
            x3_split2(r[2], r[3], pl[0].y, pl[1].y, pl[2].y);
            x3_split2(r[4], r[5], pl[0].z, pl[1].z, pl[2].z);
            x3_split2(r[6], r[7], pl[0].w, pl[1].w, pl[2].w);
            unsigned char* Bs = Wimg + ks * (3 * PJ_B_PLANE) + x3_off(col, h);
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(Bs + s * PJ_B_PLANE) = pl[s];
        }
        if (mine > 0) store_ids(0, first_ids);
    }
    __syncthreads();

    // multiplying wave w owns the tile's 64 rows x columns 32 w .. 32 w + 31: two 32 x 32 blocks
    const int mcol = w * 32 + li;
    float bv = multiplier ? bias[mcol < de ? mcol : 0] : 0.f;
    // (in a register HERE: left pending, its first use -- the epilogue, behind forty younger fetches -- became s_waitcnt vmcnt(0))
    asm volatile("" : "+v"(bv));
    auto multiply_tile = [&](int tile) {
        const int m0 = tile * PJ_TM;
        const int b_frag = x3_off(mcol, lh);
        // (the two row blocks one after the other: forty fetched pieces of the next tile are live in registers beside this.
        //  Unrolled, not a loop: in front of a loop the compiler's wait-count pass drains every outstanding load --
        //  s_waitcnt vmcnt(0) ahead of the first MFMA, i.e. no overlap at all: 55 us.)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int a_frag = x3_off(li, lh) + i * (32 * 32);
#pragma unroll
            for (int t = 0; t < PJ_MAX_KSTEPS; ++t) {
                if (t >= ksteps) break;
                const unsigned char* As = Aimg + t * (3 * PJ_A_PLANE);
                const unsigned char* Bs = Wimg + t * (3 * PJ_B_PLANE);
                x3_bf16x8 a[3], b[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    a[s] = *reinterpret_cast<const x3_bf16x8*>(As + s * PJ_A_PLANE + a_frag);
                    b[s] = *reinterpret_cast<const x3_bf16x8*>(Bs + s * PJ_B_PLANE + b_frag);
                }
                // gemm_x3's term order: the six products with p + q <= 2, smallest first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
            }
            // bias + tanh + store.  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            if (mcol < de) {
                const int rbase = m0 + i * 32 + 4 * lh;
                float* Tc = T + (size_t)rbase * de + mcol;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const float v = fast_tanh(acc[r] + bv);
                    if (rbase + dr < B) Tc[(size_t)dr * de] = v;
                }
            }
        }
    };

    // ---- the pipeline ----
    for (int i = 0; i <= mine; ++i) {
        const int tile = blockIdx.x + i * gridDim.x;          // the tile whose rows are fetched in this round (i < mine)
        const int m0 = tile * PJ_TM;
        float4 v[NI][PJ_MAX_WINDOW];
        int rows4[NI], c44[NI];
        bool have[NI];
        unsigned next_ids[2];
        if (i < mine) {
            // (1) this tile's row fetches, all of them, ids from LDS
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                have[u] = tid + u * PJ_THREADS < items;
                const int it = have[u] ? tid + u * PJ_THREADS : tid % items;
                rows4[u] = it / chunks; c44[u] = it - rows4[u] * chunks;
                const unsigned* idr = ids_lds + (i & 1) * PJ_IDS + rows4[u] * n;
#pragma unroll
                for (int q = 0; q < PJ_MAX_WINDOW; ++q) {
                    // (buffer load: descriptor in SGPRs + ONE 32-bit byte offset per fetch instead of a 64-bit address pair --
                    //  forty of those beside forty 16-byte destinations spilled; the table is below 4 GB: host check)
                    const unsigned off = idr[min(q, n - 1)] * (unsigned)dw + 4u * (unsigned)c44[u];
                    v[u][q] = tile_load16(Rw, off);
                }
            }
            // (2) ... and the ids of the tile after it
            if (i + 1 < mine) load_ids(tile + gridDim.x, next_ids);
        }
        // (3) the multiplying waves take the previous tile while the fetches fly
        if (multiplier && i >= 1) multiply_tile(tile - gridDim.x);
        __syncthreads();                                        // the A image is free
        if (i < mine) {
            // (4) pool, store h, split into the A image
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                if (!have[u]) continue;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < PJ_MAX_WINDOW; ++q)
                    if (q < n) { a.x += v[u][q].x; a.y += v[u][q].y; a.z += v[u][q].z; a.w += v[u][q].w; }
                a.x /= fn; a.y /= fn; a.z /= fn; a.w /= fn;
                const int row = rows4[u], c4 = c44[u];
                if (m0 + row < B) *reinterpret_cast<float4*>(H + (size_t)(m0 + row) * dw + 4 * c4) = a;
                else a = make_float4(0.f, 0.f, 0.f, 0.f);
                // the three planes of this piece: k step c4 / 4, quarter c4 % 4 of its 16 k (gemm_x3.h: lstore_a, k-contiguous A)
                const int ks = c4 >> 2, q = c4 & 3;
                uint2 pl[3];
                x3_split2(a.x, a.y, pl[0].x, pl[1].x, pl[2].x);
                x3_split2(a.z, a.w, pl[0].y, pl[1].y, pl[2].y);
                unsigned char* As = Aimg + ks * (3 * PJ_A_PLANE) + x3_off(row, q >> 1) + ((q & 1) << 3);
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(As + s * PJ_A_PLANE) = pl[s];
            }
            if (i + 1 < mine) store_ids((i + 1) & 1, next_ids);
        }
        __syncthreads();                                        // the A image (and the next ids) are complete
    }
}

inline bool vs_project_fused_ok(int B, int n, int dw, int de, size_t word_table_elems) {
    return word_table_elems < ((size_t)1 << 30) && n >= 1 && n <= PJ_MAX_WINDOW && dw % 16 == 0 && dw >= 16 && dw <= 128 &&
           de % 4 == 0 && de >= 4 && de <= 128 && B >= 1 && (size_t)dw * de * 4 < ((size_t)1 << 31);
}
// workgroups of the launch: one per CU, or one per tile if there are fewer tiles
inline int vs_project_grid(int B, int num_cus) { return std::max(1, std::min(cdiv(B, PJ_TM), num_cus)); }

}  // namespace sert
