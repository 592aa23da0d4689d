// Order-fixed segmented gather-reduce (replaces fp32 atomic scatter-add), gfx950.
#pragma once
#include "common.h"
#include "word_index.h"

namespace sert {

// For every item: acc = sum_{e in [begin,end)} src[row(e), :]  (e ascending),
// then  dst >= 0 : final[dst, :]    = acc / divisor  (gradient table row)
//       dst <  0 : partial[-(dst+1), :] = acc        (next level's input)
// row(e) = rows ? rows[e] / rdiv : e.  LPI lanes cooperate on one item (LPI = 32 when
// d/4 <= 32 so a wave carries two items), each lane owns float4 column chunks.
// DST_SLOT: final rows go to final[item.slot, :] (the word's rank among the batch's
// distinct words) instead of final[item.dst, :] -- loglinear per-distinct-word sums.
// touched (optional): touched[dst] = 1 for every final row written -- the optimiser
// then treats unflagged rows as zero gradient, so the table needs no memset and the
// zeros are never read back.
// LL_FINAL (loglinear distinct-word backward, with DST_SLOT): a final row is stored as
//   mask(lp) * acc - exp(lp) * rsum[slot],  lp = logp[slot, :],  mask = eps <= P <= 1-eps
// (kernels_ll.h: dZu = mask dJsum - P rsum) instead of acc / divisor.
// SKIP_DENSE (loglinear): items of the batch's dense heavy words are left out -- segsum_heavy computes
// their rows; a chunk item of such a word carries slot = -1, a final item is recognised by its slot.
struct DenseSlots {
    int n;
    int slot[kHeavyMax];
};
// Row-grouped level 0 (word_index.h: row_groups): the items are stored as eight lists; workgroup b takes the
// (b / 8)-th bundle of list b % 8 -- the dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md;
// a speed assumption only: any placement gives the same sums), so all fetches of a row range's slice of src come
// from ONE XCD's L2.  The grid is 8 x the longest list's bundle count; on = 0: items[blockIdx.x * IPB + sub].
struct XcdLists {
    int on;
    int off[8];
    int cnt[8];
    int slot_is_row;    // level 0 of a sorted index (word_index.h: BatchIndex::slot_is_row): item.slot = the item's first source row
};

template <int LPI, bool DST_SLOT, bool LL_FINAL, bool SKIP_DENSE>
__device__ __forceinline__ void segsum_rows_body(const int bx_, const int by_, const int gy_,
                                                   const float* __restrict__ src,
                                                   const int32_t* __restrict__ rows,
                                                   const int4* __restrict__ items, int nitems,
                                                   float* __restrict__ final_dst,
                                                   float* __restrict__ partial_dst, int d,
                                                   float divisor,
                                                   unsigned char* __restrict__ touched,
                                                   int rdiv,
                                                   const float* __restrict__ logp,
                                                   const float* __restrict__ rsum,
                                                   const DenseSlots& dense,
                                                   const XcdLists& xl) {
    constexpr int IPB = 256 / LPI;  // items per block
    const int sub = threadIdx.x / LPI, l = threadIdx.x % LPI;
    int item = bx_ * IPB + sub;
    if (xl.on) {
        const int x = bx_ & 7, k = (bx_ >> 3) * IPB + sub;
        if (k >= xl.cnt[x]) return;
        item = xl.off[x] + k;
    }
    if (item >= nitems) return;
    const int4 it = items[item];
    if (SKIP_DENSE) {
        bool skip = it.z < 0 && it.w < 0;
        if (it.z >= 0)
            for (int h = 0; h < dense.n; ++h) skip = skip || (dense.slot[h] == it.w);
        if (skip) return;
    }
    if (touched && l == 0 && it.z >= 0) touched[it.z] = 1;
#if defined(SERT_KO_SEG)
    if (SERT_KO_SEG == 4 && rows && it.y - it.x <= 4) return;   // 4: the short words' items leave right after their descriptor load
    if (SERT_KO_SEG == 5 && rows && it.y - it.x <= 1) return;   // 5: the singletons do
#endif
    const int chunks = d >> 2;
    // gridDim.y > 1: wide rows (d/4 > LPI) are cut into gridDim.y column groups, one
    // workgroup row each -- more, shorter chains instead of one wave walking the whole row
    // every lane of the LPI group walks the loop (the row numbers travel by lane permute, so no
    // lane may drop out): a lane whose column group is past the row computes on group 0 and
    // stores nothing
    for (int c0 = by_ * LPI; c0 < chunks; c0 += LPI * gy_) {
        const bool on = c0 + l < chunks;
        const int c = on ? c0 + l : 0;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        // LL_FINAL: the word's log-probabilities and r sum do not depend on the row sums --
        // fetch them now, in the shadow of the row loads, not after them
        float4 lp_pre = make_float4(0.f, 0.f, 0.f, 0.f);
        float rs_pre = 0.f;
        if (LL_FINAL && it.z >= 0) {
            lp_pre = *reinterpret_cast<const float4*>(logp + (size_t)it.w * d + 4 * c);
            rs_pre = rsum[it.w];
        }
        // Row numbers: one coalesced load per LPI entries (lane k holds entry k), handed round
        // with lane permutes -- not a broadcast load per entry in front of every row load.  Rows:
        // eight in flight per trip (then four, then the last one to three), so a 64-entry chunk
        // is 8 + 2 dependent memory round trips instead of 32.  Entries are added left to right
        // whatever the batching, so the sums do not depend on it.
        const int len = it.y - it.x;
        for (int base = 0; base < len; base += LPI) {
            const int cnt = min(LPI, len - base);
            int myr = it.x + base + min(l, cnt - 1);
            if (rows) {
                if (!DST_SLOT && xl.slot_is_row && len == 1) {
                    myr = it.w;              // (a one-entry item of a sorted index: its row number came with the descriptor)
                } else {
                    myr = rows[myr];
                    if (rdiv > 1) myr /= rdiv;   // source row = entry / rdiv
                }
            }
#if defined(SERT_KO_SEG)   // timing knock-outs (wrong results; tools/experiments/r05_seg_ko.sh): where do level 0's 41 us go?
            if (rows && SERT_KO_SEG == 1) myr = (it.x + base + min(l, cnt - 1)) & 65535;   // 1: rows in entry order (perfect locality)
            if (rows && SERT_KO_SEG == 3) myr = (it.x + base + min(l, cnt - 1)) & 255;     // 3: 256 distinct rows (all L1 / L2 hits)
#endif
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = __shfl(myr, k + q, LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
            if (k + 4 <= cnt) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = __shfl(myr, k + q, LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
                k += 4;
            }
            if (k < cnt) {   // one to three left: the positions past the end repeat the last entry
                float4 v[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int r = __shfl(myr, min(k + q, cnt - 1), LPI);
                    v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * d + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (k + q < cnt) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
        }
        if (!on) continue;
#if defined(SERT_KO_SEG)
        if (SERT_KO_SEG == 2 && rows && a.x != 12345.678f) continue;                             // 2: no stores
#endif
        if (it.z >= 0 && LL_FINAL) {
            const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
            const size_t o = (size_t)it.w * d + 4 * c;
            const float4 lp = lp_pre;
            const float rs = rs_pre;
            a.x = ((lp.x >= LOGLO && lp.x <= LOGHI) ? a.x : 0.f) - __expf(lp.x) * rs;
            a.y = ((lp.y >= LOGLO && lp.y <= LOGHI) ? a.y : 0.f) - __expf(lp.y) * rs;
            a.z = ((lp.z >= LOGLO && lp.z <= LOGHI) ? a.z : 0.f) - __expf(lp.z) * rs;
            a.w = ((lp.w >= LOGLO && lp.w <= LOGHI) ? a.w : 0.f) - __expf(lp.w) * rs;
            *reinterpret_cast<float4*>(final_dst + o) = a;
        } else if (it.z >= 0) {
            a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
            *reinterpret_cast<float4*>(final_dst + (size_t)(DST_SLOT ? it.w : it.z) * d + 4 * c) = a;
        } else {
            *reinterpret_cast<float4*>(partial_dst + (size_t)(-(it.z + 1)) * d + 4 * c) = a;
        }
    }
}

template <int LPI, bool DST_SLOT = false, bool LL_FINAL = false, bool SKIP_DENSE = false>
__global__ __launch_bounds__(256) void segsum_rows(const float* __restrict__ src,
                                                   const int32_t* __restrict__ rows,
                                                   const int4* __restrict__ items, int nitems,
                                                   float* __restrict__ final_dst,
                                                   float* __restrict__ partial_dst, int d,
                                                   float divisor,
                                                   unsigned char* __restrict__ touched,
                                                   int rdiv = 1,
                                                   const float* __restrict__ logp = nullptr,
                                                   const float* __restrict__ rsum = nullptr,
                                                   const DenseSlots dense = DenseSlots(),
                                                   const XcdLists xl = XcdLists()) {
    segsum_rows_body<LPI, DST_SLOT, LL_FINAL, SKIP_DENSE>((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, src, rows, items, nitems,
                                                          final_dst, partial_dst, d, divisor, touched, rdiv, logp, rsum, dense, xl);
}

// Scalar variant for d % 4 != 0 (and for scalars, d = 1): one wave per (item, 64-column group),
// gridDim.y column groups; four entries in flight per trip.  Same summation order as one entry
// at a time (left to right), so the result does not depend on the unrolling.
// LL (loglinear, odd V_e; kernels_ll.h: dZu = mask dJsum - P rsum): every wave also sums its item's scalars r_ik
// (rsrc / rrows: the same item list over the per-occurrence scalars -- one entry per lane and wave_sum, exactly
// segsum_scalar_wave's additions), column group 0 stores that sum, and a word's FINAL row is stored as
//   mask(lp) * acc - exp(lp) * rsum,  lp = logp[slot, :]
// -- the expression of ll_dzu_combine -- so that neither the two scalar launches nor the combine launch exist
// (W3C settings, 715 experts: three launches of ~5 us each in an 16-launch chain).
template <bool DST_SLOT = false, bool LL = false>
__global__ __launch_bounds__(256) void segsum_rows_scalar(const float* __restrict__ src,
                                                          const int32_t* __restrict__ rows,
                                                          const int4* __restrict__ items,
                                                          int nitems, float* __restrict__ final_dst,
                                                          float* __restrict__ partial_dst, int d,
                                                          float divisor,
                                                          unsigned char* __restrict__ touched,
                                                          int rdiv = 1,
                                                          const float* __restrict__ logp = nullptr,
                                                          const float* __restrict__ rsrc = nullptr,
                                                          const int32_t* __restrict__ rrows = nullptr,
                                                          float* __restrict__ rsum = nullptr,
                                                          float* __restrict__ rpart = nullptr) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int4 it = items[item];
    if (touched && lane == 0 && blockIdx.y == 0 && it.z >= 0) touched[it.z] = 1;
    float rs = 0.f;
    if (LL) {
        for (int e0 = it.x; e0 < it.y; e0 += 64) {      // (one trip: chunks hold <= 64 entries)
            const int e = e0 + lane;
            float v = 0.f;
            if (e < it.y) v = rsrc[rrows ? rrows[e] : e];
            rs += wave_sum(v);
        }
        if (lane == 0 && blockIdx.y == 0) {
            if (it.z >= 0) rsum[it.w] = rs;
            else rpart[-(it.z + 1)] = rs;
        }
    }
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    for (int c = blockIdx.y * 64 + lane; c < d; c += 64 * gridDim.y) {
        float a = 0.f;
        int e = it.x;
        for (; e + 4 <= it.y; e += 4) {
            int r0, r1, r2, r3;
            if (rows) { r0 = rows[e]; r1 = rows[e + 1]; r2 = rows[e + 2]; r3 = rows[e + 3]; }
            else      { r0 = e; r1 = e + 1; r2 = e + 2; r3 = e + 3; }
            if (rows && rdiv > 1) { r0 /= rdiv; r1 /= rdiv; r2 /= rdiv; r3 /= rdiv; }
            const float v0 = src[(size_t)r0 * d + c], v1 = src[(size_t)r1 * d + c];
            const float v2 = src[(size_t)r2 * d + c], v3 = src[(size_t)r3 * d + c];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; e < it.y; ++e) {
            const int r = rows ? (rdiv > 1 ? rows[e] / rdiv : rows[e]) : e;
            a += src[(size_t)r * d + c];
        }
        if (it.z >= 0) {
            const size_t t = (size_t)(DST_SLOT ? it.w : it.z) * d + c;
            if (LL) {
                const float lp = logp[t];
                const float dj = (lp >= LOGLO && lp <= LOGHI) ? a : 0.f;
                final_dst[t] = dj - __expf(lp) * rs;
            } else {
                final_dst[t] = a / divisor;
            }
        } else partial_dst[(size_t)(-(it.z + 1)) * d + c] = a;
    }
}

// d = 1 (the per-position scalars r_ik of the loglinear backward, summed per word): one wave per item,
// one ENTRY PER LANE (an item has at most kSegChunk = 64 entries), a wave reduction in a fixed lane order.
// segsum_rows_scalar walked the entries four at a time on ONE active lane -- sixteen dependent trips for a
// full chunk: 39 us for the 44 k items of a C2-dims batch, two dependent loads here.
__global__ __launch_bounds__(256) void segsum_scalar_wave(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                          const int4* __restrict__ items, int nitems,
                                                          float* __restrict__ final_dst, float* __restrict__ partial_dst) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int4 it = items[item];
    float a = 0.f;
    for (int e0 = it.x; e0 < it.y; e0 += 64) {      // (one trip: chunks hold <= 64 entries)
        const int e = e0 + lane;
        float v = 0.f;
        if (e < it.y) v = src[rows ? rows[e] : e];
        a += wave_sum(v);
    }
    if (lane == 0) {
        if (it.z >= 0) final_dst[it.w] = a;        // (DST_SLOT: the word's rank among the batch's distinct words)
        else partial_dst[-(it.z + 1)] = a;
    }
}

// Levels 1 and 2 of a three-level tree in ONE launch (word_index.h: heavy_off).  Workgroups
// [0, nb_normal) take 32 level-1 items each and store the final ones (a chunk item -- dst < 0 -- is
// left to its word's workgroup); workgroup nb_normal + h sums the <= 32 chunk items of heavy word h,
// one lane group each, and adds the chunk sums in chunk order out of LDS: exactly the additions the
// separate level-2 launch makes, in the same order (bit-identical), without its ~6 us of launch.
// (Used when no word of the batch has more than 32 chunk items = 131 072 occurrences; the bench's
// synthetic batches, whose clipped Zipf tail piles 30 % of the tokens on one word, keep the two launches.)
__global__ __launch_bounds__(1024) void segsum_upper_fused(const float* __restrict__ src,
                                                           const int4* __restrict__ items, int nitems,
                                                           int nb_normal, const int4* __restrict__ heavy,
                                                           float* __restrict__ final_dst, int d, float divisor) {
    __shared__ float4 red[64][32];
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int chunks = d >> 2;               // 32 float4 columns per blockIdx.y (d = 300: three column slabs)
    const bool on = (int)blockIdx.y * 32 + l < chunks;
    const int c = on ? (int)blockIdx.y * 32 + l : 0;
    auto sum_rows = [&](int lo, int hi) -> float4 {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = lo;
        for (; e + 8 <= hi; e += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(e + q) * d + 4 * c);
#pragma unroll
            for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        }
        for (; e < hi; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)e * d + 4 * c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        return a;
    };
    if ((int)blockIdx.x < nb_normal) {
        const int item = blockIdx.x * 32 + sub;
        if (item >= nitems) return;
        const int4 it = items[item];
        if (it.z < 0) return;
        float4 a = sum_rows(it.x, it.y);
        a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
        if (on) *reinterpret_cast<float4*>(final_dst + (size_t)it.z * d + 4 * c) = a;
        return;
    }
    const int4 h = heavy[blockIdx.x - nb_normal];     // {first chunk item, chunk items, word, -}
    for (int q = sub; q < h.y; q += 32) {
        const int4 it = items[h.x + q];
        red[q][l] = sum_rows(it.x, it.y);
    }
    __syncthreads();
    if (sub == 0 && on) {
        float4 s = red[0][l];
        for (int q = 1; q < h.y; ++q) {
            const float4 x = red[q][l];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        s.x /= divisor; s.y /= divisor; s.z /= divisor; s.w /= divisor;
        *reinterpret_cast<float4*>(final_dst + (size_t)h.z * d + 4 * c) = s;
    }
}

// ---- dense heavy words (word_index.h: kHeavyMax) -------------------------------------------------
// part[block][h][cols] = sum over the block's kHeavyRowsPerBlock batch rows i of cnt[i][h] * src[i, cols],
// for the batch's kHeavyMax heavy words at once: ONE coalesced streaming pass over src (B x d) instead of
// one row fetch per occurrence.  1024 threads = 32 lane groups of 32 lanes; a group takes eight rows of
// the block (all eight in flight), a lane one float4 column of the 128-column slab blockIdx.y.  The
// groups are combined in a fixed order: the two halves of a wave by a lane exchange, then the sixteen
// waves through FOUR LDS slots (32 KB: wave w adds into slot w % 4 in round w / 4), so that two
// workgroups fit a CU and one's reduction overlaps the other's loads (one 128 KB slot per wave: 100 us
// for the 262 MB of dJ at C2 dims -- a workgroup alone on its CU loads a third of the time).
__global__ __launch_bounds__(1024) void segsum_heavy(const float* __restrict__ src, const uint4* __restrict__ cnt16,
                                                     int B, int d, float* __restrict__ part) {
    extern __shared__ float4 hv_lds[];   // [4 slots][kHeavyMax][32]
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5, wv = threadIdx.x >> 6;
    const int d4 = d >> 2;
    // 1-D grid, column slab FASTEST: the workgroups running at the same time cover whole rows of src (the
    // slabs of a row block are neighbours in launch order), not one 512-byte piece of every row
    const int nslab = (d4 + 31) / 32;
    const int slab = blockIdx.x % nslab, rblk = blockIdx.x / nslab;
    const int ch = slab * 32 + l;                  // this lane's float4 column
    const bool on = ch < d4;
    // (two-wide vector type: the 4 x 16 x 8 multiply-adds per thread compile to v_pk_fma_f32, two per
    //  instruction -- the pass is VALU-bound otherwise: 2.1 GFLOP on the scalar FMA rate is 35 us)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[kHeavyMax][2];
#pragma unroll
    for (int h = 0; h < kHeavyMax; ++h) { acc2[h][0] = (f32x2)(0.f); acc2[h][1] = (f32x2)(0.f); }
    constexpr int RPG = kHeavyRowsPerBlock / 32;   // rows per lane group
    const int row0 = rblk * kHeavyRowsPerBlock + g * RPG;
    uint4 c[RPG];
    float4 v[RPG];
#pragma unroll
    for (int q = 0; q < RPG; ++q) {
        const int i = min(row0 + q, B - 1);
        c[q] = cnt16[i];
        if (row0 + q >= B) c[q] = make_uint4(0u, 0u, 0u, 0u);
        v[q] = on ? *reinterpret_cast<const float4*>(src + (size_t)i * d + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < RPG; ++q) {
        const unsigned cw[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
        f32x2 lo, hi;
        lo.x = v[q].x; lo.y = v[q].y; hi.x = v[q].z; hi.y = v[q].w;
#pragma unroll
        for (int h = 0; h < kHeavyMax; ++h) {
            const float f = (float)((cw[h >> 2] >> (8 * (h & 3))) & 0xffu);
            const f32x2 ff = (f32x2)(f);
            acc2[h][0] = __builtin_elementwise_fma(ff, lo, acc2[h][0]);
            acc2[h][1] = __builtin_elementwise_fma(ff, hi, acc2[h][1]);
        }
    }
    float4 acc[kHeavyMax];
#pragma unroll
    for (int h = 0; h < kHeavyMax; ++h) acc[h] = make_float4(acc2[h][0].x, acc2[h][0].y, acc2[h][1].x, acc2[h][1].y);
    // lower half of the wave += upper half (group 2w + group 2w+1), then one slot per wave in LDS
#pragma unroll
    for (int h = 0; h < kHeavyMax; ++h) {
        acc[h].x += __shfl_xor(acc[h].x, 32); acc[h].y += __shfl_xor(acc[h].y, 32);
        acc[h].z += __shfl_xor(acc[h].z, 32); acc[h].w += __shfl_xor(acc[h].w, 32);
    }
    for (int round = 0; round < 4; ++round) {
        if ((wv >> 2) == round && (threadIdx.x & 63) < 32) {
            float4* slot = hv_lds + (size_t)(wv & 3) * kHeavyMax * 32;
#pragma unroll
            for (int h = 0; h < kHeavyMax; ++h) {
                if (round == 0) slot[h * 32 + l] = acc[h];
                else { float4 o = slot[h * 32 + l]; o.x += acc[h].x; o.y += acc[h].y; o.z += acc[h].z; o.w += acc[h].w; slot[h * 32 + l] = o; }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < kHeavyMax * 32) {
        const int h = threadIdx.x >> 5;
        float4 a = hv_lds[h * 32 + l];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 t = hv_lds[(w * kHeavyMax + h) * 32 + l];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        if (on) reinterpret_cast<float4*>(part)[((size_t)rblk * kHeavyMax + h) * d4 + ch] = a;
    }
}

// Wide rows (the loglinear dJ: V_e floats): one WAVE per batch row and 512-column slab, so that a row's
// counts are wave-uniform -- most of a row's sixteen counts are zero (ten tokens, three to four distinct
// heavy words), and a uniform branch skips their multiply-adds: the pass was VALU-bound on 16 x V_e x B
// products of which under a quarter are non-zero.  512 threads = 8 waves x 32 rows (four trips of eight
// rows in flight); waves combined through two LDS slots in four rounds (fixed order).
__global__ __launch_bounds__(512) void segsum_heavy_wide(const float* __restrict__ src, const uint4* __restrict__ cnt16,
                                                         int B, int d, float* __restrict__ part) {
    extern __shared__ float4 hv_lds[];   // [2 slots][kHeavyMax][128]
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int d4 = d >> 2;
    const int nslab = (d4 + 127) / 128;
    const int slab = blockIdx.x % nslab, rblk = blockIdx.x / nslab;
    const int ch0 = slab * 128 + lane, ch1 = ch0 + 64;
    const bool on0 = ch0 < d4, on1 = ch1 < d4;
    f32x2 acc[kHeavyMax][4];
#pragma unroll
    for (int h = 0; h < kHeavyMax; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[h][q] = (f32x2)(0.f);
    constexpr int RPW = kHeavyRowsPerBlock / 8;    // rows per wave
    const int row0 = rblk * kHeavyRowsPerBlock + wv * RPW;
    for (int t0 = 0; t0 < RPW; t0 += 8) {
        uint4 c[8];
        float4 v0[8], v1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = __builtin_amdgcn_readfirstlane(min(row0 + t0 + q, B - 1));
            c[q] = cnt16[i];
            if (row0 + t0 + q >= B) c[q] = make_uint4(0u, 0u, 0u, 0u);
            v0[q] = on0 ? *reinterpret_cast<const float4*>(src + (size_t)i * d + 4 * ch0) : make_float4(0.f, 0.f, 0.f, 0.f);
            v1[q] = on1 ? *reinterpret_cast<const float4*>(src + (size_t)i * d + 4 * ch1) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned cw[4] = {(unsigned)__builtin_amdgcn_readfirstlane((int)c[q].x), (unsigned)__builtin_amdgcn_readfirstlane((int)c[q].y),
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)c[q].z), (unsigned)__builtin_amdgcn_readfirstlane((int)c[q].w)};
            f32x2 a, b, cc, dd;
            a.x = v0[q].x; a.y = v0[q].y; b.x = v0[q].z; b.y = v0[q].w;
            cc.x = v1[q].x; cc.y = v1[q].y; dd.x = v1[q].z; dd.y = v1[q].w;
#pragma unroll
            for (int h = 0; h < kHeavyMax; ++h) {
                const unsigned byte = (cw[h >> 2] >> (8 * (h & 3))) & 0xffu;
                if (byte != 0u) {                       // (wave-uniform)
                    const f32x2 ff = (f32x2)((float)byte);
                    acc[h][0] = __builtin_elementwise_fma(ff, a, acc[h][0]);
                    acc[h][1] = __builtin_elementwise_fma(ff, b, acc[h][1]);
                    acc[h][2] = __builtin_elementwise_fma(ff, cc, acc[h][2]);
                    acc[h][3] = __builtin_elementwise_fma(ff, dd, acc[h][3]);
                }
            }
        }
    }
    for (int round = 0; round < 4; ++round) {
        if ((wv >> 1) == round) {
            float4* slot = hv_lds + (size_t)(wv & 1) * kHeavyMax * 128;
#pragma unroll
            for (int h = 0; h < kHeavyMax; ++h) {
                float4 x0 = make_float4(acc[h][0].x, acc[h][0].y, acc[h][1].x, acc[h][1].y);
                float4 x1 = make_float4(acc[h][2].x, acc[h][2].y, acc[h][3].x, acc[h][3].y);
                if (round != 0) {
                    const float4 o0 = slot[h * 128 + lane], o1 = slot[h * 128 + 64 + lane];
                    x0.x += o0.x; x0.y += o0.y; x0.z += o0.z; x0.w += o0.w;
                    x1.x += o1.x; x1.y += o1.y; x1.z += o1.z; x1.w += o1.w;
                }
                slot[h * 128 + lane] = x0;
                slot[h * 128 + 64 + lane] = x1;
            }
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < kHeavyMax * 128; k += 512) {
        const int h = k >> 7, cl = k & 127;
        const int ch = slab * 128 + cl;
        if (ch >= d4) continue;
        const float4 p0 = hv_lds[k], p1 = hv_lds[kHeavyMax * 128 + k];
        reinterpret_cast<float4*>(part)[((size_t)rblk * kHeavyMax + h) * d4 + ch] =
            make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
    }
}

// final[word[h], :] = (sum over the blocks, in block order within eight interleaved groups, then the
// groups in order) / divisor.  One workgroup per heavy word.
__global__ __launch_bounds__(256) void segsum_heavy_combine(const float* __restrict__ part, int nblocks, int d,
                                                            const int32_t* __restrict__ words, int nheavy,
                                                            float* __restrict__ final_dst, float divisor) {
    __shared__ float4 hc_lds[8][32];
    const int h = blockIdx.x;
    if (h >= nheavy) return;
    const int d4 = d >> 2;
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = blockIdx.y * 32 + l;          // (gridDim.y slabs of 32 float4 columns)
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < d4) {
#pragma unroll 8
        for (int b = g; b < nblocks; b += 8) {
            const float4 v = p4[((size_t)b * kHeavyMax + h) * d4 + ch];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    hc_lds[g][l] = a;
    __syncthreads();
    if (g == 0 && ch < d4) {
        a = hc_lds[0][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 v = hc_lds[q][l]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
        reinterpret_cast<float4*>(final_dst)[(size_t)words[h] * d4 + ch] = a;
    }
}

// ---- the dense heavy words INSIDE the tree's launches (vectorspace word gradient, round 5) -------------------------------
// segsum_heavy + segsum_heavy_combine in front of the tree cost what the heavy words' entries saved (round 3: 56.4 against
// 55.6 us at C2): a single wave of 256 large workgroups bound by its one dependent chain, then another launch.  But the pass
// is a STREAM (address-computable loads, no descriptor -> row numbers -> rows chain) and the tree's level 0 is bound by
// exactly those chains -- side by side they use different things.  segsum_rows_plus is the tree's launch with `extra`
// leading workgroups (x) that do the other job:
//   kind 1 (beside level 0): the count-weighted partial sums of PlusJob::rpb batch rows for the batch's <= 16 dense words.
//       256 threads = 4 waves; waves 0, 1 take words 0-7, waves 2, 3 words 8-15 (eight float4 accumulators per lane: the
//       launch keeps the tree's eight waves per SIMD); the four lane groups of a word half take every fourth row, four rows in
//       flight per trip; their sums meet in the order (g0 + g1) + (g2 + g3) (lane permute, then 8 kB of LDS) and leave as
//       part[row block][word][:] -- the layout of segsum_heavy, over smaller row blocks.
//   kind 2 (beside level 1): segsum_heavy_combine's body, one workgroup per dense word (x) and 32-column slab (y).
// The column slab of an extra workgroup is blockIdx.y: the launch must have gridDim.y == cdiv(d / 4, 32) (the LPI = 32 forms).
// Batch rows per extra workgroup of kind 1 (PlusJob::rpb, a multiple of 16): about one workgroup per CU -- fewer, longer ones leave
// fewer partial rows to the combine (a chain of its own beside level 1), too few make the stream outlast level 0.  Measured
// (tools/experiments/r05_heavy_rows.sh; 64 / 128 / 256 / 512 rows): C2 0.2398 / 0.2364 / 0.2353 / 0.2360 ms, C2 dims at 8192 rows
// 0.0949 / 0.0961 / 0.1024 / 0.1166, C4 1.344 / 1.336 / 1.317 / 1.333, loglinear at batch 65536 0.923 / 0.904 / 0.893 / 1.02.
constexpr int kHeavyRowsFusedMin = 64, kHeavyRowsFusedMax = 256;
inline int heavy_rows_fused(int B) {
    int r = kHeavyRowsFusedMin;
    while (r < kHeavyRowsFusedMax && B / (2 * r) >= 256) r *= 2;
    return r;
}
struct PlusJob {
    int kind;                 // 0: none, 1: heavy partial sums, 2: combine
    int extra;                // leading workgroups (x) that do it
    const float* src;         // kind 1: the source matrix (B x d);  kind 2: the partials
    const uint4* cnt16;       // kind 1: per batch row, kHeavyMax occurrence counts (bytes)
    float* part;              // kind 1: [extra][kHeavyMax][d]
    const int32_t* words;     // kind 2: the dense words' table rows
    int nheavy;               // kind 2
    int nblocks;              // kind 2: row blocks of the partials
    int B;
    int slot_is_row;          // the TREE's items: see XcdLists
    int rpb;                  // kind 1: batch rows per extra workgroup
};

__device__ __forceinline__ void heavy_rows_body(const int rblk, const int slab, const PlusJob& job, int d, float4 (*lds)[8][32]) {
    const int l = threadIdx.x & 31, wv = threadIdx.x >> 6, half = wv >> 1;
    const int rs = (wv & 1) * 2 + ((threadIdx.x >> 5) & 1);    // which fourth of the rows
    const int d4 = d >> 2, ch = slab * 32 + l;
    const bool on = ch < d4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[8][2];
#pragma unroll
    for (int h = 0; h < 8; ++h) { acc2[h][0] = (f32x2)(0.f); acc2[h][1] = (f32x2)(0.f); }
    const int row0 = rblk * job.rpb + rs;
    const uint2* cnt8 = reinterpret_cast<const uint2*>(job.cnt16) + half;   // (this half's eight count bytes of row i: cnt8[2 i])
#pragma unroll 1
    for (int t = 0; t < job.rpb / 16; ++t) {
        uint2 c[4];
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = row0 + (t * 4 + q) * 4;
            const int i = min(r, job.B - 1);
            c[q] = cnt8[2 * (size_t)i];
            if (r >= job.B) c[q] = make_uint2(0u, 0u);
            v[q] = on ? *reinterpret_cast<const float4*>(job.src + (size_t)i * d + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned cw[2] = {c[q].x, c[q].y};
            f32x2 lo, hi;
            lo.x = v[q].x; lo.y = v[q].y; hi.x = v[q].z; hi.y = v[q].w;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const f32x2 ff = (f32x2)((float)((cw[h >> 2] >> (8 * (h & 3))) & 0xffu));
                acc2[h][0] = __builtin_elementwise_fma(ff, lo, acc2[h][0]);
                acc2[h][1] = __builtin_elementwise_fma(ff, hi, acc2[h][1]);
            }
        }
    }
    float4 acc[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        acc[h] = make_float4(acc2[h][0].x, acc2[h][0].y, acc2[h][1].x, acc2[h][1].y);
        // lower lane group of the wave += the upper one
        acc[h].x += __shfl_xor(acc[h].x, 32); acc[h].y += __shfl_xor(acc[h].y, 32);
        acc[h].z += __shfl_xor(acc[h].z, 32); acc[h].w += __shfl_xor(acc[h].w, 32);
    }
    if ((wv & 1) && (threadIdx.x & 63) < 32) {
#pragma unroll
        for (int h = 0; h < 8; ++h) lds[half][h][l] = acc[h];
    }
    __syncthreads();
    if (!(wv & 1) && (threadIdx.x & 63) < 32 && on) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float4 o = lds[half][h][l];
            float4 a = acc[h];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            reinterpret_cast<float4*>(job.part)[((size_t)rblk * kHeavyMax + half * 8 + h) * d4 + ch] = a;
        }
    }
}

__device__ __forceinline__ void heavy_combine_body(const int h, const int slab, const PlusJob& job, int d, float* __restrict__ final_dst,
                                                   float divisor, float4 (*lds)[32]) {
    if (h >= job.nheavy) return;
    const int d4 = d >> 2;
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = slab * 32 + l;
    const float4* p4 = reinterpret_cast<const float4*>(job.src);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < d4) {
#pragma unroll 8
        for (int b = g; b < job.nblocks; b += 8) {
            const float4 v = p4[((size_t)b * kHeavyMax + h) * d4 + ch];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    lds[g][l] = a;
    __syncthreads();
    if (g == 0 && ch < d4) {
        a = lds[0][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 v = lds[q][l]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        a.x /= divisor; a.y /= divisor; a.z /= divisor; a.w /= divisor;
        reinterpret_cast<float4*>(final_dst)[(size_t)job.words[h] * d4 + ch] = a;
    }
}

__global__ __launch_bounds__(256) void segsum_rows_plus(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                        const int4* __restrict__ items, int nitems, float* __restrict__ final_dst,
                                                        float* __restrict__ partial_dst, int d, float divisor, const PlusJob job) {
    __shared__ float4 plus_lds[2][8][32];   // 8 kB (the tree's workgroups do not touch it)
    if ((int)blockIdx.x < job.extra) {
        if (job.kind == 1) heavy_rows_body((int)blockIdx.x, (int)blockIdx.y, job, d, plus_lds);
        else heavy_combine_body((int)blockIdx.x, (int)blockIdx.y, job, d, final_dst, divisor, plus_lds[0]);
        return;
    }
    XcdLists xl = XcdLists();
    xl.slot_is_row = job.slot_is_row;
    segsum_rows_body<32, false, false, false>((int)blockIdx.x - job.extra, (int)blockIdx.y, (int)gridDim.y, src, rows, items, nitems,
                                              final_dst, partial_dst, d, divisor, nullptr, 1, nullptr, nullptr, DenseSlots(), xl);
}

// ... and the loglinear per-word dZ sums (V_e-wide rows, 64-lane items, gridDim.y = cdiv(V_e / 4, 64) column groups): the same
// two jobs beside segsum_rows<64, true, true, true>.  kind 1: a wave per (word half, every second row) -- a row's counts are
// wave-uniform, the zero ones are skipped by a scalar branch (a row of ten tokens holds three or four distinct heavy words);
// the two row subsets of a word half meet in 16 kB of LDS.  kind 2: segsum_heavy_combine_ll's finishing expression
// (mask dJsum - P rsum), one workgroup per dense word and 64-column group, the row blocks' partials split over its four waves.
struct PlusJobLL {
    PlusJob j;
    DenseSlots dense;         // kind 2: the dense words' ranks among the batch's distinct words (= their rows of dZu)
    const float* logp;
    const float* rsum;
};

__device__ __forceinline__ void heavy_rows_body64(const int rblk, const int slab, const PlusJob& job, int d, float4 (*lds)[8][64]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = wv >> 1, rs = wv & 1;
    const int d4 = d >> 2, ch = slab * 64 + lane;
    const bool on = ch < d4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[8][2];
#pragma unroll
    for (int h = 0; h < 8; ++h) { acc2[h][0] = (f32x2)(0.f); acc2[h][1] = (f32x2)(0.f); }
    const int row0 = rblk * job.rpb + rs;
    const uint2* cnt8 = reinterpret_cast<const uint2*>(job.cnt16) + half;
#pragma unroll 1
    for (int t = 0; t < job.rpb / 8; ++t) {
        uint2 c[4];
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = row0 + (t * 4 + q) * 2;
            const int i = __builtin_amdgcn_readfirstlane(min(r, job.B - 1));
            c[q] = cnt8[2 * (size_t)i];
            if (r >= job.B) c[q] = make_uint2(0u, 0u);
            v[q] = on ? *reinterpret_cast<const float4*>(job.src + (size_t)i * d + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned cw[2] = {(unsigned)__builtin_amdgcn_readfirstlane((int)c[q].x), (unsigned)__builtin_amdgcn_readfirstlane((int)c[q].y)};
            f32x2 lo, hi;
            lo.x = v[q].x; lo.y = v[q].y; hi.x = v[q].z; hi.y = v[q].w;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const unsigned byte = (cw[h >> 2] >> (8 * (h & 3))) & 0xffu;
                if (byte != 0u) {                       // (wave-uniform)
                    const f32x2 ff = (f32x2)((float)byte);
                    acc2[h][0] = __builtin_elementwise_fma(ff, lo, acc2[h][0]);
                    acc2[h][1] = __builtin_elementwise_fma(ff, hi, acc2[h][1]);
                }
            }
        }
    }
    if (rs) {
#pragma unroll
        for (int h = 0; h < 8; ++h) lds[half][h][lane] = make_float4(acc2[h][0].x, acc2[h][0].y, acc2[h][1].x, acc2[h][1].y);
    }
    __syncthreads();
    if (!rs && on) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float4 o = lds[half][h][lane];
            reinterpret_cast<float4*>(job.part)[((size_t)rblk * kHeavyMax + half * 8 + h) * d4 + ch] =
                make_float4(acc2[h][0].x + o.x, acc2[h][0].y + o.y, acc2[h][1].x + o.z, acc2[h][1].y + o.w);
        }
    }
}

__device__ __forceinline__ void heavy_combine_ll_body64(const int h, const int slab, const PlusJobLL& job, int d,
                                                        float* __restrict__ final_dst, float4 (*lds)[64]) {
    if (h >= job.j.nheavy) return;
    const int d4 = d >> 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = slab * 64 + lane;
    const float4* p4 = reinterpret_cast<const float4*>(job.j.src);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < d4) {
#pragma unroll 8
        for (int b = wv; b < job.j.nblocks; b += 4) {
            const float4 v = p4[((size_t)b * kHeavyMax + h) * d4 + ch];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    lds[wv][lane] = a;
    __syncthreads();
    if (wv == 0 && ch < d4) {
        a = lds[0][lane];
#pragma unroll
        for (int q = 1; q < 4; ++q) { const float4 v = lds[q][lane]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        const int slot = job.dense.slot[h];
        const float rs = job.rsum[slot];
        const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
        const size_t o = (size_t)slot * d + 4 * ch;
        const float4 lp = *reinterpret_cast<const float4*>(job.logp + o);
        a.x = ((lp.x >= LOGLO && lp.x <= LOGHI) ? a.x : 0.f) - __expf(lp.x) * rs;
        a.y = ((lp.y >= LOGLO && lp.y <= LOGHI) ? a.y : 0.f) - __expf(lp.y) * rs;
        a.z = ((lp.z >= LOGLO && lp.z <= LOGHI) ? a.z : 0.f) - __expf(lp.z) * rs;
        a.w = ((lp.w >= LOGLO && lp.w <= LOGHI) ? a.w : 0.f) - __expf(lp.w) * rs;
        *reinterpret_cast<float4*>(final_dst + o) = a;
    }
}

__global__ __launch_bounds__(256) void segsum_rows_plus_ll(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                           const int4* __restrict__ items, int nitems, float* __restrict__ final_dst,
                                                           float* __restrict__ partial_dst, int d, const DenseSlots dense,
                                                           const PlusJobLL job) {
    __shared__ float4 plus_lds[2][8][64];   // 16 kB (the tree's workgroups do not touch it)
    if ((int)blockIdx.x < job.j.extra) {
        if (job.j.kind == 1) heavy_rows_body64((int)blockIdx.x, (int)blockIdx.y, job.j, d, plus_lds);
        else heavy_combine_ll_body64((int)blockIdx.x, (int)blockIdx.y, job, d, final_dst, plus_lds[0]);
        return;
    }
    segsum_rows_body<64, true, true, true>((int)blockIdx.x - job.j.extra, (int)blockIdx.y, (int)gridDim.y, src, rows, items, nitems,
                                           final_dst, partial_dst, d, 1.0f, nullptr, 1, job.logp, job.rsum, dense, XcdLists());
}

// Loglinear: row `slot[h]` of dZu = mask(lp) * (sum of the heavy word's dJ rows) - exp(lp) * rsum[slot]
// (the LL_FINAL store of segsum_rows), from the per-block partials of segsum_heavy over dJ.
__global__ __launch_bounds__(256) void segsum_heavy_combine_ll(const float* __restrict__ part, int nblocks, int d,
                                                               const DenseSlots dense, float* __restrict__ final_dst,
                                                               const float* __restrict__ logp, const float* __restrict__ rsum) {
    __shared__ float4 hc_lds[8][32];
    const int h = blockIdx.x;
    if (h >= dense.n) return;
    const int d4 = d >> 2;
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ch = blockIdx.y * 32 + l;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < d4) {
#pragma unroll 8
        for (int b = g; b < nblocks; b += 8) {
            const float4 v = p4[((size_t)b * kHeavyMax + h) * d4 + ch];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    hc_lds[g][l] = a;
    __syncthreads();
    const int slot = dense.slot[h];
    const float rs = rsum[slot];
    const float LOGLO = logf(SERT_CLIP_LO), LOGHI = logf(SERT_CLIP_HI);
    if (g == 0 && ch < d4) {
        a = hc_lds[0][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 v = hc_lds[q][l]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        const size_t o = (size_t)slot * d + 4 * ch;
        const float4 lp = *reinterpret_cast<const float4*>(logp + o);
        a.x = ((lp.x >= LOGLO && lp.x <= LOGHI) ? a.x : 0.f) - __expf(lp.x) * rs;
        a.y = ((lp.y >= LOGLO && lp.y <= LOGHI) ? a.y : 0.f) - __expf(lp.y) * rs;
        a.z = ((lp.z >= LOGLO && lp.z <= LOGHI) ? a.z : 0.f) - __expf(lp.z) * rs;
        a.w = ((lp.w >= LOGLO && lp.w <= LOGHI) ? a.w : 0.f) - __expf(lp.w) * rs;
        *reinterpret_cast<float4*>(final_dst + o) = a;
    }
}

}  // namespace sert
