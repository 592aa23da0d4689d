// The one-launch entity gradient for few pairs over a mid-size table (egrad_ranges): measured faster ALONE and equal in the STEP
// (profiles/r05_experiments.txt; sert_hip.hip: egrad_ranges_ok) -- compiled into a -DSERT_VARIANTS build only since round 6.
#pragma once
#include "../kernels_egrad.h"

namespace sert {

// ---- few pairs, mid-size entity table: ONE launch, no sort ------------------------------------------------------
// The reference's product-search settings (batch 4096, z = 10: 45 056 (pair, entity) keys over 32 768 entities) ran the general
// path above: a stable counting sort in two digits (6 launches), the chunked reduce and the carry fix-up -- eight dependent
// launches of 5-15 us each on the side stream, 150 us beside the word table's update, longer than the rest of the step (the
// next step's loss kernel waited for its end: round-5 timeline).  For that regime one workgroup per RANGE of kERange entities
//   (a) scans ALL candidate ids (P x 4 bytes out of L2, coalesced) and appends the indices of those in its range to a
//       list in LDS (unordered: atomic append),
//   (b) sorts the list by (entity, pair index) -- a bitonic sort of a few hundred 32-bit keys in LDS --,
//   (c) walks it: lane group g of 32 lanes takes the range's entities g, g + 8, ..., finds the entity's run by binary search and
//       adds coef . clip(t_row) over it IN PAIR ORDER (the order the stable sort above produces; the association differs
//       from the chunked reduce: one chain per entity instead of chunks + carries), then stores the row -- zeros for an
//       entity without a pair, so every row of dR_e is written, as the general path does.
// A range whose list would exceed kERList entries (an adversarial concentration of labels) takes the slow walk (d):
// each lane group scans the candidates in order for its entities.  Deterministic either way.
// Applies when ranges x P stays small (the scan is ranges x P id reads): sert_hip.hip: egrad_ranges_ok.
constexpr int kERange = 32;       // entities per workgroup (128: one workgroup per CU at the product-search table, one wave per SIMD -- every phase latency-bound, 44 us)
constexpr int kERList = 8192;     // list capacity (32 kB of LDS)

__global__ __launch_bounds__(256) void egrad_ranges(const int32_t* __restrict__ cand, const float* __restrict__ coef,
                                                    const float* __restrict__ T, int P, int zp1, int de, int V,
                                                    float* __restrict__ GRe) {
    __shared__ unsigned keys[kERList];
    __shared__ unsigned s_n;
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * kERange;
    const int er = min(kERange, V - e0);
    if (tid == 0) s_n = 0u;
    __syncthreads();
    // (a) the candidates of this range: key = local entity << 20 | pair index  (P <= 2^20: host check)
    // (sixteen ids per thread and trip, the next trip's fetched while this one is tested: four per trip with the test right
    //  behind the loads was one exposed L2 round trip per 1024 candidates -- 44 of them at the product-search settings, 45 us)
    constexpr int NU = 16;
    int c[NU], cn[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { const int i = u * 256 + tid; c[u] = i < P ? cand[i] : -1; }
    for (int base = 0; base < P; base += 256 * NU) {
        const int nbase = base + 256 * NU;
#pragma unroll
        for (int u = 0; u < NU; ++u) { const int i = nbase + u * 256 + tid; cn[u] = i < P ? cand[i] : -1; }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const unsigned le = (unsigned)(c[u] - e0);
            if (c[u] >= 0 && le < (unsigned)er) {
                const unsigned slot = atomicAdd(&s_n, 1u);
                if (slot < (unsigned)kERList) keys[slot] = (le << 20) | (unsigned)(base + u * 256 + tid);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) c[u] = cn[u];
    }
    __syncthreads();
    const unsigned n = s_n;
    const int g = tid >> 5, l = tid & 31;       // lane group, lane: float4 column pieces l, l + 32, ...
    const int pieces = de >> 2;                 // de % 4 == 0 (host check)
    if (n <= (unsigned)kERList) {
        // (b) bitonic sort of the keys, ascending, padded with ~0
        unsigned sn = 2;
        while (sn < n) sn <<= 1;
        for (unsigned i = n + tid; i < sn; i += 256) keys[i] = 0xffffffffu;
        __syncthreads();
        for (unsigned size = 2; size <= sn; size <<= 1) {
            for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                for (unsigned i = tid; i < sn / 2; i += 256) {
                    const unsigned lo = 2 * i - (i & (stride - 1));
                    const unsigned hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const unsigned a = keys[lo], b = keys[hi];
                    if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
                }
                __syncthreads();
            }
        }
        // (c) one chain per entity, in pair order.  Lane group g owns the CONTIGUOUS block of entities [g er/8, (g+1) er/8): its
        // entries are one slice of the sorted list, walked eight entries per trip -- their coefficients and projection rows are
        // fetched together (an entity has 1.4 pairs on average: one chain of dependent loads per entity took 57 us) -- and added
        // entry by entry; an entity's row is stored when the walk leaves it, zeros for an entity without a pair.
        const int per = (er + 7) / 8;
        const int le_lo = min(er, g * per), le_hi = min(er, le_lo + per);
        auto lower = [&](unsigned want) {
            unsigned lo = 0, hi = n;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if (keys[mid] < want) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        const unsigned k_lo = lower((unsigned)le_lo << 20), k_hi = lower((unsigned)le_hi << 20);
        for (int p0 = 0; p0 < pieces; p0 += 32) {
            const int cpc = p0 + l;
            const bool on = cpc < pieces;
            int cur = le_lo;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            auto store_cur = [&]() {
                if (on) *reinterpret_cast<float4*>(GRe + (size_t)(e0 + cur) * de + 4 * cpc) = acc;
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
                ++cur;
            };
            for (unsigned k0 = k_lo; k0 < k_hi; k0 += 8) {
                unsigned kk[8];
                float cf[8];
                float4 t4[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    kk[q] = keys[min(k0 + q, k_hi - 1)];
                    const int pr = (int)(kk[q] & 0xfffffu);
                    cf[q] = coef[pr];
                    t4[q] = on ? *reinterpret_cast<const float4*>(T + (size_t)(pr / zp1) * de + 4 * cpc) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (k0 + q >= k_hi) break;
                    const int le = (int)(kk[q] >> 20);
                    while (cur < le) store_cur();        // (rows of the entities the walk passed: their sums, or zeros)
                    acc.x += cf[q] * fminf(fmaxf(t4[q].x, -SERT_CLIP_HI), SERT_CLIP_HI);
                    acc.y += cf[q] * fminf(fmaxf(t4[q].y, -SERT_CLIP_HI), SERT_CLIP_HI);
                    acc.z += cf[q] * fminf(fmaxf(t4[q].z, -SERT_CLIP_HI), SERT_CLIP_HI);
                    acc.w += cf[q] * fminf(fmaxf(t4[q].w, -SERT_CLIP_HI), SERT_CLIP_HI);
                }
            }
            while (cur < le_hi) store_cur();
        }
        return;
    }
    // (d) more pairs in this range than the list holds: every lane group walks ALL candidates, in order, for its entities
    for (int le = g; le < er; le += 8) {
        for (int p0 = 0; p0 < pieces; p0 += 32) {
            const int cpc = p0 + l;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int base = 0; base < P; base += 32) {
                const int i = base + l;
                const bool hit = i < P && cand[i] == e0 + le;
                unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                unsigned mm = (tid & 32) ? (unsigned)(mask >> 32) : (unsigned)mask;     // this lane group's half of the wave
                while (mm) {
                    const int b = __builtin_ctz(mm);
                    mm &= mm - 1;
                    const int pr = base + b;
                    const float cf = coef[pr];
                    if (cpc < pieces) {
                        const float4 t4 = *reinterpret_cast<const float4*>(T + (size_t)(pr / zp1) * de + 4 * cpc);
                        acc.x += cf * fminf(fmaxf(t4.x, -SERT_CLIP_HI), SERT_CLIP_HI);
                        acc.y += cf * fminf(fmaxf(t4.y, -SERT_CLIP_HI), SERT_CLIP_HI);
                        acc.z += cf * fminf(fmaxf(t4.z, -SERT_CLIP_HI), SERT_CLIP_HI);
                        acc.w += cf * fminf(fmaxf(t4.w, -SERT_CLIP_HI), SERT_CLIP_HI);
                    }
                }
            }
            if (cpc < pieces) *reinterpret_cast<float4*>(GRe + (size_t)(e0 + le) * de + 4 * cpc) = acc;
        }
    }
}

}  // namespace sert
