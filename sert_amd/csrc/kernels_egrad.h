// Entity-table gradient of the NCE loss without atomics, gfx950.
//
// dR_e[e] = sum over all (row i, candidate j) with c_ij = e of du_ij * clip(t_i)
// (autodiff of sert/models.py:990 + :897; Theano AdvancedIncSubtensor1).
// The B*(1+z) (entity, pair) keys are counting-sorted by entity (kernels_sort.h; stable, so the
// order inside an entity is the pair order), then reduced in fixed 16-pair
// chunks; runs that cross a chunk boundary leave per-chunk carries that a
// second kernel adds in chunk order.  Same association every run => the result
// is deterministic, and no address is ever contended.
#pragma once
#include "common.h"

namespace sert {

constexpr int kEChunk = 16;   // pairs per chunk = lanes per chunk group

// 16 lanes per chunk of 16 consecutive sorted pairs (4 chunks per wave).  Lane l
// of the group preloads pair l (key, source row, coefficient) and owns the
// VEC-wide column pieces l, l+16, ... of the accumulated rows.  NCH pieces per
// lane per pass; when the row is wider than 16*NCH pieces the chunk is walked
// again for the next column block (only the scalar fallback needs that).
template <int VEC, int NCH>
__global__ __launch_bounds__(256) void egrad_chunk_reduce(
    const int32_t* __restrict__ keys, const int32_t* __restrict__ pairs,
    const float* __restrict__ coef, const float* __restrict__ T, int total, int zp1, int de,
    float* __restrict__ GRe, float* __restrict__ head, float* __restrict__ tail,
    int32_t* __restrict__ run_start, int32_t* __restrict__ run_end) {
    const int l = threadIdx.x & 15;
    const int gbase = (threadIdx.x & 63) & ~15;  // first lane of this group inside the wave
    const int chunk = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int cb = chunk * kEChunk;
    if (cb >= total) return;
    const int ce = min(total, cb + kEChunk);
    const int cnt = ce - cb;
    const int pieces = de / VEC;
    int my_key = -1, my_row = 0;
    float my_coef = 0.f;
    if (l < cnt) {
        my_key = keys[cb + l];
        const int pr = pairs[cb + l];
        my_coef = coef[pr];
        my_row = pr / zp1;
    }
    const int prev_key = (cb > 0) ? keys[cb - 1] : -1;
    const int next_key = (ce < total) ? keys[ce] : -1;
    const int first_key = __shfl(my_key, gbase, kWave);

    for (int p0 = 0; p0 < pieces; p0 += 16 * NCH) {
        float acc[NCH][VEC];
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
        int cur = first_key;
        bool at_begin = true;
        int jb = 0;

        auto flush = [&](int e, bool touches_begin, bool touches_end, int jb, int je) {
            // sorted positions of the entity's run: recorded by the chunk that
            // holds its first / last pair (fix-up then needs no search)
            if (p0 == 0 && l == 0) {
                if (!touches_begin) run_start[e] = cb + jb;
                if (!touches_end) run_end[e] = cb + je;
            }
            float* dst;
            if (!touches_begin && !touches_end) dst = GRe + (size_t)e * de;   // sole owner
            else if (touches_end) dst = tail + (size_t)chunk * de;
            else dst = head + (size_t)chunk * de;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = p0 + l + 16 * q;
                if (c < pieces) {
                    if (VEC == 4)
                        *reinterpret_cast<float4*>(dst + 4 * c) =
                            make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
                    else
                        dst[c] = acc[q][0];
                }
            }
        };

        for (int j = 0; j < cnt; ++j) {
            const int k = __shfl(my_key, gbase + j, kWave);
            const int row = __shfl(my_row, gbase + j, kWave);
            const float cf = __shfl(my_coef, gbase + j, kWave);
            if (k != cur) {
                flush(cur, at_begin && prev_key == cur, false, jb, j);
                jb = j;
#pragma unroll
                for (int q = 0; q < NCH; ++q)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
                cur = k;
                at_begin = false;
            }
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = p0 + l + 16 * q;
                if (c < pieces) {
                    float tv[VEC];
                    if (VEC == 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(T + (size_t)row * de + 4 * c);
                        tv[0] = t4.x; tv[1 % VEC] = t4.y; tv[2 % VEC] = t4.z; tv[3 % VEC] = t4.w;
                    } else {
                        tv[0] = T[(size_t)row * de + c];
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[q][v] += cf * fminf(fmaxf(tv[v], -SERT_CLIP_HI), SERT_CLIP_HI);
                }
            }
        }
        flush(cur, at_begin && prev_key == cur, next_key == cur, jb, cnt);
    }
}

// One wave per entity: add the carries of the chunks its run [s,t) spans.  The
// four 16-lane rows of the wave take every 4th chunk (independent load chains),
// then combine in a fixed order.
template <int VEC>
__global__ __launch_bounds__(256) void egrad_fixup(const int32_t* __restrict__ run_start,
                                                   const int32_t* __restrict__ run_end, int V,
                                                   int de, const float* __restrict__ head,
                                                   const float* __restrict__ tail,
                                                   float* __restrict__ GRe) {
    const int lane = threadIdx.x & 63;
    const int l = lane & 15, g = lane >> 4;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= V) return;
    const int s = run_start[e], t = run_end[e];
    const int pieces = de / VEC;
    if (t <= s) {                            // entity not in this batch (the bounds are zeroed): a zero row
        if (g == 0)                          // (dR_e itself is NOT zeroed per step: every row is written)
            for (int c = l; c < pieces; c += 16)
#pragma unroll
                for (int v = 0; v < VEC; ++v) GRe[(size_t)e * de + VEC * c + v] = 0.f;
        return;
    }
    const int cs = s / kEChunk, cl = (t - 1) / kEChunk;
    if (cs == cl) return;                    // run inside one chunk: written directly
    for (int c = l; c < pieces; c += 16) {
        float a[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = 0.f;
        // chunks cs .. cl-1 carry the run in their tail slot, chunk cl in its head slot
#pragma unroll 4
        for (int k = cs + g; k < cl; k += 4) {
            if (VEC == 4) {
                const float4 v4 = *reinterpret_cast<const float4*>(tail + (size_t)k * de + 4 * c);
                a[0] += v4.x; a[1 % VEC] += v4.y; a[2 % VEC] += v4.z; a[3 % VEC] += v4.w;
            } else {
                a[0] += tail[(size_t)k * de + c];
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float x = a[v];
            x += __shfl_xor(x, 16, kWave);
            x += __shfl_xor(x, 32, kWave);
            a[v] = x + head[(size_t)cl * de + VEC * c + v];
        }
        if (g == 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) GRe[(size_t)e * de + VEC * c + v] = a[v];
        }
    }
}

// Few entities (V < 256): a run spans thousands of chunks and one wave per entity leaves the
// chip idle (10 entities at batch 65536: 198 us).  One workgroup per entity instead: its sixteen
// 16-lane rows take every 16th chunk, the row sums meet in LDS and are added in row order.
template <int VEC>
__global__ __launch_bounds__(256) void egrad_fixup_wg(const int32_t* __restrict__ run_start,
                                                      const int32_t* __restrict__ run_end, int V,
                                                      int de, const float* __restrict__ head,
                                                      const float* __restrict__ tail,
                                                      float* __restrict__ GRe) {
    __shared__ float part[16][512];
    const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x;
    if (e >= V) return;
    const int s = run_start[e], t = run_end[e];
    const int pieces = de / VEC;
    if (t <= s) {                            // workgroup-uniform; entity not in this batch: a zero row
        for (int c = threadIdx.x; c < de; c += 256) GRe[(size_t)e * de + c] = 0.f;
        return;
    }
    const int cs = s / kEChunk, cl = (t - 1) / kEChunk;
    if (cs == cl) return;
    for (int c = l; c < pieces; c += 16) {
        float a[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = 0.f;
#pragma unroll 4
        for (int k = cs + g; k < cl; k += 16) {
            if (VEC == 4) {
                const float4 v4 = *reinterpret_cast<const float4*>(tail + (size_t)k * de + 4 * c);
                a[0] += v4.x; a[1 % VEC] += v4.y; a[2 % VEC] += v4.z; a[3 % VEC] += v4.w;
            } else {
                a[0] += tail[(size_t)k * de + c];
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[g][VEC * c + v] = a[v];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < de; c += 256) {
        float x = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) x += part[r][c];
        GRe[(size_t)e * de + c] = x + head[(size_t)cl * de + c];
    }
}

// ================================================================================================
// Small entity vocabularies (V_e <= 2048, d_e <= 256): the same gradient WITHOUT the global sort.
//
// The sorted path above costs three latency-bound sort launches, a reduction that fetches every
// row of T eleven times from wherever it happens to live (B*(1+z) random 512-byte rows out of a
// 33 MB table: 399 MB of fabric traffic at C2, the largest item of the whole step) and a fix-up:
// 109 us serial, 56 us of the 369 us step even beside the main stream (knock-out measurement).
// Here the entity vocabulary is cut into RANGES of 16 entities whose accumulators fit a wave's registers,
// and the batch into row GROUPS whose slice of T fits one XCD's L2 (<= 2 MB):
//
//   egrad_bucket   one workgroup per SUB-group of <= 256 rows: its (1+z)*rows pairs (cand is
//                  row-major, so they are contiguous) are partitioned by entity range inside the
//                  workgroup -- a stable LDS counting sort on <= 128 bins, ranks by ballot matching
//                  in program order, so positions are a pure function of the input.
//                  -> bucket entries (pair << 4 | entity - range base) and per-(sub-group, range) offsets
//   egrad_acc      workgroup (group g, range r): every wave owns a fixed set of the group's
//                  sub-groups and walks their lists for range r in order, fetching coef and the T
//                  rows sixteen pairs at a time and adding coef * clip(t) into per-entity accumulators
//                  held in registers; the four waves' accumulators are then added in wave order.
//                  All workgroups of a group sit on ONE XCD (blockIdx % 8 is the XCD in practice --
//                  a performance assumption only), so ten of the eleven reads of a T row are L2 hits.
//                  -> partial[g][e][:]   (G * V_e * d_e floats: 8 MB at C2)
//   egrad_group_sum  dR_e[e] = sum_g partial[g][e]  in group order.
//
// Order-fixed, no atomics on the accumulation.  Fabric traffic ~ |T| + 2 |partial| instead of
// (1+z) |T|.  Beyond V_e = 2048 the per-range lists get too short to amortise a workgroup; larger
// vocabularies keep the sorted path.
#ifndef SERT_EL_BATCH
#define SERT_EL_BATCH 16
#endif
#ifndef SERT_EL_DB
#define SERT_EL_DB 0               /* 1: two batches in flight (more registers, fewer resident waves) */
#endif
constexpr int kElBatch = SERT_EL_BATCH;   // pairs (T rows) in flight per wave
constexpr int kElSubPairs = 4096;  // pair capacity of one sub-group (16 per thread)
constexpr int kElMaxRanges = 128;

__device__ __forceinline__ float4 clip4(float4 t) {
    t.x = fminf(fmaxf(t.x, -SERT_CLIP_HI), SERT_CLIP_HI);
    t.y = fminf(fmaxf(t.y, -SERT_CLIP_HI), SERT_CLIP_HI);
    t.z = fminf(fmaxf(t.z, -SERT_CLIP_HI), SERT_CLIP_HI);
    t.w = fminf(fmaxf(t.w, -SERT_CLIP_HI), SERT_CLIP_HI);
    return t;
}

// er_shift: er = 1 << er_shift entities per range (<= 16).  sub_rows rows per sub-group,
// sub_rows * c1 <= kElSubPairs.  entries: (B*c1) ints, offs: (num_sub, num_ranges + 1) ints.
// y / neg != nullptr: the keys straight from the labels and the negatives -- cand[i, j] = j ? neg[i, j - 1] : y[i] is what the
// loss kernel writes (kernels_vs.h) -- so that the partition can run BEFORE the loss kernel, beside the forward (round 6).
__global__ __launch_bounds__(512) void egrad_bucket(const int32_t* __restrict__ cand, int B, int c1, int sub_rows,
                                                    int er_shift, int num_ranges, int32_t* __restrict__ entries,
                                                    int32_t* __restrict__ offs, const int32_t* __restrict__ y = nullptr,
                                                    const int32_t* __restrict__ neg = nullptr) {
    constexpr int NW = 8;                        // waves per workgroup
    __shared__ int32_t wh[NW][kElMaxRanges];    // per-wave range counts -> start positions
    __shared__ int32_t base[kElMaxRanges + 1];
    const int sg = blockIdx.x;
    const int row_lo = sg * sub_rows, row_hi = min(B, row_lo + sub_rows);
    const int p_lo = row_lo * c1, npairs = (row_hi - row_lo) * c1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int b = tid; b < NW * kElMaxRanges; b += 64 * NW) (&wh[0][0])[b] = 0;
    __syncthreads();
    // every wave owns a contiguous share of the pairs (a multiple of 64 long); its keys are
    // fetched up front (16 independent loads per lane: the phases below never wait on memory)
    const int per_wave = ((npairs + NW - 1) / NW + 63) & ~63;
    const int w_lo = min(npairs, w * per_wave), w_hi = min(npairs, w_lo + per_wave);
    constexpr int KPL = kElSubPairs / (64 * NW) + 1;      // keys per lane (per_wave <= 4096/8 rounded up to 64)
    int key[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int i = w_lo + k * 64 + lane;
        const int p = p_lo + min(i, max(npairs - 1, 0));
        if (y) {
            const int row = p / c1, j = p - row * c1;
            key[k] = j ? neg[(size_t)row * (c1 - 1) + (j - 1)] : y[row];
        } else {
            key[k] = cand[p];
        }
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile int32_t* mine = wh[w];
    int nbits = 0;
    while ((1 << nbits) < num_ranges) ++nbits;
    // phase A: counts
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int i = w_lo + k * 64 + lane;
        const bool active = i < w_hi;
        const int rid = key[k] >> er_shift;
        unsigned long long peers = __ballot(active);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (rid >> bit) & 1;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        if (active && (peers & lt_mask) == 0) mine[rid] += __popcll(peers);   // leader of its range in this step
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // phase B: range bases = exclusive scan over the <= 128 range totals (wave 0: lane r owns
    // ranges r and r + 64), then the waves' cursors
    if (w == 0) {
        const int r1 = lane + 64;
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            if (lane < num_ranges) t0 += wh[ww][lane];
            if (r1 < num_ranges) t1 += wh[ww][r1];
        }
        int inc0 = t0, inc1 = t1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o0 = __shfl_up(inc0, off, kWave), o1 = __shfl_up(inc1, off, kWave);
            if (lane >= off) { inc0 += o0; inc1 += o1; }
        }
        const int total0 = __shfl(inc0, 63, kWave);
        if (lane < num_ranges) base[lane] = inc0 - t0;
        if (r1 < num_ranges) base[r1] = total0 + inc1 - t1;
        if (lane == 63) base[num_ranges] = total0 + inc1;
    }
    __syncthreads();
    for (int r = tid; r <= num_ranges; r += 64 * NW) offs[(size_t)sg * (num_ranges + 1) + r] = base[r];
    for (int r = tid; r < num_ranges; r += 64 * NW) {
        int g = base[r];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { const int c = wh[ww][r]; wh[ww][r] = g; g += c; }
    }
    __syncthreads();
    // phase C: stable placement
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int i = w_lo + k * 64 + lane;
        const bool active = i < w_hi;
        const int rid = key[k] >> er_shift;
        unsigned long long peers = __ballot(active);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (rid >> bit) & 1;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        if (active) {
            const int pos = mine[rid] + __popcll(peers & lt_mask);
            entries[p_lo + pos] = ((p_lo + i) << 4) | (key[k] & ((1 << er_shift) - 1));
        }
        __builtin_amdgcn_wave_barrier();
        if (active && (peers & lt_mask) == 0) mine[rid] += __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
}

// Workgroup (group g, entity range r of 16 entities), four waves.  Wave w walks the range-r lists of
// sub-groups w, w + 4, ... of the group in order.  The accumulators of the 16 entities live in
// REGISTERS (VPL floats per lane and entity: lane l owns columns [VPL l, VPL l + VPL) of a row; the
// entity of a pair is wave-uniform, so the accumulator is picked by a scalar register index, no LDS
// and no conflicts), which leaves the CU free to host 16+ waves that hide each other's dependent
// loads: per list one coalesced load of the entries, one gather of their coefficients, then the T
// rows sixteen at a time.  The four waves' accumulators meet in LDS and are added in wave order.
template <int VPL>
__global__ __launch_bounds__(256) void egrad_acc(const int32_t* __restrict__ entries, const int32_t* __restrict__ offs,
                                                 const float* __restrict__ coef, const float* __restrict__ T, int c1,
                                                 int de, int V, int sub_rows, int num_sub, int subs_per_group,
                                                 int num_groups, int num_ranges, float* __restrict__ partial) {
    extern __shared__ float4 el_smem[];
    float* red = reinterpret_cast<float*>(el_smem);                 // [4 waves][16 entities][de]
    // workgroup -> (group, range); the groups of XCD x are x, x + 8, ...
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = xcd + 8 * (j / num_ranges), r = j % num_ranges;
    if (g >= num_groups) return;
    const int e0 = r << 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane * VPL;
    const bool col_ok = col < de;
    float acc[16][VPL];
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[e][v] = 0.f;

    // T rows of batch `b0` of the current chunk into `tv` (no use of the values: no wait)
    auto fetch = [&](float (&tv)[kElBatch][VPL], int rowv, int b0, int cnt) {
#pragma unroll
        for (int q = 0; q < kElBatch; ++q) {
            const int row = __builtin_amdgcn_readlane(rowv, min(b0 + q, cnt - 1));
            const float* trow = T + (size_t)row * de + (col_ok ? col : 0);
            if (VPL == 2) {
                const float2 t2 = *reinterpret_cast<const float2*>(trow);
                tv[q][0] = t2.x; tv[q][1 % VPL] = t2.y;
            } else {
                const float4 t4 = *reinterpret_cast<const float4*>(trow);
                tv[q][0] = t4.x; tv[q][1 % VPL] = t4.y; tv[q][2 % VPL] = t4.z; tv[q][3 % VPL] = t4.w;
            }
        }
    };
    auto accumulate = [&](const float (&tv)[kElBatch][VPL], int en, float cfv, int b0, int cnt) {
        const int nb = min(kElBatch, cnt - b0);
#pragma unroll
        for (int q = 0; q < kElBatch; ++q) {
            if (q < nb) {
                const int el = __builtin_amdgcn_readlane(en, b0 + q) & 15;       // wave-uniform
                const float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfv), b0 + q));
#pragma unroll
                for (int v = 0; v < VPL; ++v)
                    acc[el][v] += cf * fminf(fmaxf(tv[q][v], -SERT_CLIP_HI), SERT_CLIP_HI);
            }
        }
    };

    // the (lo, hi) of this wave's lists, all fetched in one go
    constexpr int kMaxLists = 8;              // lists per wave handled in one pass
    for (int s0 = w; s0 < subs_per_group; s0 += 4 * kMaxLists) {
        int lo[kMaxLists], hi[kMaxLists];
#pragma unroll
        for (int k = 0; k < kMaxLists; ++k) {
            const int sg = g * subs_per_group + s0 + 4 * k;
            const bool ok = (s0 + 4 * k < subs_per_group) && sg < num_sub;
            const int32_t* o = offs + (size_t)(ok ? sg : 0) * (num_ranges + 1) + r;
            const int a = o[0], b = o[1];
            lo[k] = ok ? a : 0;
            hi[k] = ok ? b : 0;
        }
#pragma unroll
        for (int k = 0; k < kMaxLists; ++k) {
            const int l0 = __builtin_amdgcn_readfirstlane(lo[k]), h0 = __builtin_amdgcn_readfirstlane(hi[k]);
            const int sg = g * subs_per_group + s0 + 4 * k;
            const int32_t* ent = entries + (size_t)sg * sub_rows * c1;
            for (int base = l0; base < h0; base += 64) {
                const int cnt = min(64, h0 - base);
                const int en = ent[base + min(lane, cnt - 1)];         // one coalesced load: 64 entries
                const int pidx = en >> 4;
                const float cfv = coef[pidx];                          // gather of their coefficients
                const int rowv = pidx / c1;
#if SERT_EL_DB
                // T rows kElBatch at a time, the next batch in flight while this one is added
                float tva[kElBatch][VPL], tvb[kElBatch][VPL];
                fetch(tva, rowv, 0, cnt);
                for (int b0 = 0; b0 < cnt; b0 += 2 * kElBatch) {
                    if (b0 + kElBatch < cnt) fetch(tvb, rowv, b0 + kElBatch, cnt);
                    accumulate(tva, en, cfv, b0, cnt);
                    if (b0 + kElBatch < cnt) {
                        if (b0 + 2 * kElBatch < cnt) fetch(tva, rowv, b0 + 2 * kElBatch, cnt);
                        accumulate(tvb, en, cfv, b0 + kElBatch, cnt);
                    }
                }
#else
                // T rows kElBatch at a time (one buffer: 16+ resident waves per CU hide the hops of
                // one another better than two batches in flight per wave do -- 43 vs 53 us at C2)
                for (int b0 = 0; b0 < cnt; b0 += kElBatch) {
                    float tv[kElBatch][VPL];
                    fetch(tv, rowv, b0, cnt);
                    accumulate(tv, en, cfv, b0, cnt);
                }
#endif
            }
        }
    }
    // the four waves' accumulators, added in wave order
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (col_ok) red[((size_t)w * 16 + e) * de + col + v] = acc[e][v];
    __syncthreads();
    const int de4 = de >> 2;
    for (int i = threadIdx.x; i < 16 * de4; i += 256) {
        const int e = i / de4, c = i - e * de4;
        if (e0 + e >= V) continue;
        const float4* rp = reinterpret_cast<const float4*>(red);
        float4 s4 = rp[(size_t)e * de4 + c];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 x = rp[((size_t)k * 16 + e) * de4 + c];
            s4.x += x.x; s4.y += x.y; s4.z += x.z; s4.w += x.w;
        }
        reinterpret_cast<float4*>(partial + ((size_t)g * V + e0 + e) * de)[c] = s4;
    }
}

// dR_e = sum over the row groups of their partial tables, in group order.
__global__ __launch_bounds__(256) void egrad_group_sum(const float* __restrict__ partial, int num_groups,
                                                       size_t table4, float* __restrict__ GRe) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < table4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<const float4*>(partial)[i];
        for (int g = 1; g < num_groups; ++g) {
            const float4 x = reinterpret_cast<const float4*>(partial)[(size_t)g * table4 + i];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        reinterpret_cast<float4*>(GRe)[i] = s;
    }
}

}  // namespace sert
