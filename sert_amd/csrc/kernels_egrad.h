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
    if (t <= s) return;                      // entity not in this batch (arrays are zeroed)
    const int cs = s / kEChunk, cl = (t - 1) / kEChunk;
    if (cs == cl) return;                    // run inside one chunk: written directly
    const int pieces = de / VEC;
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
    if (t <= s) return;                      // workgroup-uniform
    const int cs = s / kEChunk, cl = (t - 1) / kEChunk;
    if (cs == cl) return;
    const int pieces = de / VEC;
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

}  // namespace sert
