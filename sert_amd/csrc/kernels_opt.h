// Dense optimiser + L2 + loss reduction kernels, gfx950.  Pure HBM streaming.
//
// The reference updates EVERY element of EVERY table EVERY step: the L2 term
// lambda/(2B)*||p||^2 (sert/models.py:764-795) gives every parameter a non-zero
// gradient, and lasagne.updates.adam / adadelta (models.py:922 / :820, applied
// :548-549) are dense element-wise maps.  One fused kernel per tensor:
//   g  = grad + (lambda/B) * p          (b: no L2 [upstream: regularizable=False])
//   sumsq += p^2  (pre-update, for the loss the step returns)
//   state/param update
#pragma once
#include "common.h"

namespace sert {

#ifndef SERT_OPT_BLOCKS
#define SERT_OPT_BLOCKS 2048   // (1024 / 4096 / 8192 measured within noise of 2048 at C2 and C4)
#endif
constexpr int kOptBlocks = SERT_OPT_BLOCKS;  // fixed grid => fixed reduction tree => deterministic

struct AdamArgs {
    float l2k;   // lambda / B
    float a_t;   // lr*sqrt(1-b2^t)/(1-b1^t), evaluated on the host in fp32
    float b1, b2, eps;
};

// The element updates below are compiled with contraction OFF (#pragma clang fp contract(off): every product and sum
// rounded on its own; HIP's __fmul_rn / __fadd_rn are plain operators and do not prevent it): the same element is
// updated by different kernels -- the dense launches, the lazy launch and its catch-up loop, the small-tensor launch, the
// fused tail -- and must come out with the same bits whichever one it was (round 4: b2 v + (1 - b2) g g was contracted
// differently in two of them, 1 ulp apart).  The order is NumPy's on the oracle's expressions
// (the CPU restatement the tests compare with: its Adam / Adadelta updates); only the square root and the division go through the
// hardware's 1-ulp instructions (see there).  sum(p^2) accumulates through an explicit fma everywhere.
__device__ __forceinline__ float sq_acc(float acc, float x) { return __builtin_fmaf(x, x, acc); }

// Lasagne 0.1 adam [upstream]:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p = p - a_t*m/(sqrt(v)+eps)
__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, const AdamArgs& a,
                                          float omb1, float omb2, float& ss, float& ss_new) {
#pragma clang fp contract(off)
    const float pv = p;
    const float l2 = a.l2k * pv;
    const float gv = g + l2;
    ss = sq_acc(ss, pv);
    const float m1 = a.b1 * m, m2 = omb1 * gv;
    const float mv = m1 + m2;
    const float v1 = a.b2 * v, v2 = (omb2 * gv) * gv;
    const float vv = v1 + v2;
    m = mv;
    v = vv;
    // sqrt and the division through the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 instead of the IEEE expansions (~10
    // instructions each): irrelevant to the memory-bound dense launch, but the lazy launch's catch-up loop is VALU-bound
    // (round 4: W3C loglinear 331 -> 263 us per step with them, 288 without).  <= 2 ulp on the update term, i.e. ~2e-10
    // of a parameter per step against a parity tolerance of 1e-4.  -DSERT_OPT_IEEE_DIV restores sqrtf and '/'.
#ifndef SERT_OPT_IEEE_DIV
    const float num = a.a_t * mv, den = __builtin_amdgcn_sqrtf(vv) + a.eps;
    const float quo = num * __builtin_amdgcn_rcpf(den);
    const float pn = pv - quo;
#else
    const float num = a.a_t * mv, den = sqrtf(vv) + a.eps;
    const float pn = pv - num / den;
#endif
    p = pn;
    ss_new = sq_acc(ss_new, pn);    // (the same expression, in the same element order, as `ss` of the NEXT step's launch)
    g = gv;
}

// Row filter of the streaming optimiser launches (word table, single GPU): `bits` holds one bit
// per table row, set iff a token of the current batch points to the row (built once per batch at
// upload, word_index.h).  The gradient of an unset row is zero and is NOT read (the table is
// never zeroed either); kRowsAll updates every row, kRowsTouched / kRowsUntouched only the rows
// whose bit is set / clear -- the untouched rows (56 % of the table at C2) are not read by the
// batch's own forward, so their update runs beside it instead of behind the backward.
enum { kRowsAll = 0, kRowsTouched = 1, kRowsUntouched = 2 };
__device__ __forceinline__ bool row_bit(const uint32_t* __restrict__ bits, unsigned row) {
    return (bits[row >> 5] >> (row & 31)) & 1u;
}

// 16-byte accesses (4 streams in, 3-4 out); tensors are 16-byte aligned, the
// (< 4 element) tail is handled by the first threads of block 0.
// bits/row_len: see above (row_len % 4 == 0, count < 2^32).
template <bool STORE_G>
__global__ __launch_bounds__(256) void adam_l2(float* __restrict__ p, float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v,
                                               size_t count, AdamArgs a,
                                               float* __restrict__ sumsq_partial,
                                               const uint32_t* __restrict__ bits = nullptr,
                                               unsigned row_len = 1, int rows_mode = kRowsAll,
                                               float* __restrict__ sumsq_new_partial = nullptr) {
    // sumsq_new_partial: also leave the sums of squares of the UPDATED values, partitioned exactly as
    // sumsq_partial is -- what the next step's launch over the same tensor would compute as its pre-update
    // sums, bit for bit (the deferred entity-table update of the side-heavy schedule, sert_hip.hip)
    __shared__ float red[4];
    float ss = 0.f, ssn = 0.f;
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
    const size_t n4 = count >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    // each workgroup streams ONE contiguous slice (a grid-stride walk at a 4 MB
    // power-of-two stride keeps all workgroups on the same HBM channels at once)
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const bool hit = !bits || row_bit(bits, ((unsigned)i << 2) / row_len);
        if (rows_mode != kRowsAll && hit != (rows_mode == kRowsTouched)) continue;
#ifndef SERT_ADAM_NO_NT
        // the optimiser state is touched once per step and by nobody else: streaming (nt) accesses keep
        // it out of the way of the tables the gathers live on (59.5 -> 58.5 us, step -1.5 % at C2)
        typedef float nt_f4 __attribute__((ext_vector_type(4)));
        float4 pp = p4[i];
        const nt_f4 mr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(m) + i);
        const nt_f4 vr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(v) + i);
        float4 mm = make_float4(mr.x, mr.y, mr.z, mr.w), vv = make_float4(vr.x, vr.y, vr.z, vr.w);
#else
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
#endif
        float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hit) gg = g4[i];
        adam_elem(pp.x, gg.x, mm.x, vv.x, a, omb1, omb2, ss, ssn);
        adam_elem(pp.y, gg.y, mm.y, vv.y, a, omb1, omb2, ss, ssn);
        adam_elem(pp.z, gg.z, mm.z, vv.z, a, omb1, omb2, ss, ssn);
        adam_elem(pp.w, gg.w, mm.w, vv.w, a, omb1, omb2, ss, ssn);
#ifndef SERT_ADAM_NO_NT
        p4[i] = pp;
        { nt_f4 t; t.x = mm.x; t.y = mm.y; t.z = mm.z; t.w = mm.w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(m) + i); }
        { nt_f4 t; t.x = vv.x; t.y = vv.y; t.z = vv.z; t.w = vv.w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(v) + i); }
#else
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
#endif
        if (STORE_G) g4[i] = gg;
    }
    if (blockIdx.x == 0 && !bits) {   // (< 4 element tail; a row-filtered table has none)
        const size_t i = (n4 << 2) + threadIdx.x;
        if (i < count) {
            float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
            adam_elem(pp, gg, mm, vv, a, omb1, omb2, ss, ssn);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (STORE_G) g[i] = gg;
        }
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
    if (sumsq_new_partial) {      // (kernel-uniform)
        __syncthreads();
        const float tn = block_sum_256(ssn, red);
        if (threadIdx.x == 0) sumsq_new_partial[blockIdx.x] = tn;
    }
}

// The pre-update sums of squares of adam_l2 WITHOUT the update: same slices, same element order, same
// block reduction -- partial[b] equals what adam_l2<...>(p, ..., sumsq_partial) with the same grid leaves
// in sumsq_partial[b] (first step of a deferred entity-table update, or after the host replaced the table).
__global__ __launch_bounds__(256) void sumsq_like_adam(const float* __restrict__ p, size_t count,
                                                       float* __restrict__ sumsq_partial) {
    __shared__ float red[4];
    float ss = 0.f;
    const size_t n4 = count >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float4 pp = p4[i];
        ss = sq_acc(ss, pp.x);     // (the expression of adam_elem)
        ss = sq_acc(ss, pp.y);
        ss = sq_acc(ss, pp.z);
        ss = sq_acc(ss, pp.w);
    }
    if (blockIdx.x == 0) {
        const size_t i = (n4 << 2) + threadIdx.x;
        if (i < count) ss = sq_acc(ss, p[i]);
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
}

struct AdadeltaArgs {
    float l2k, lr, rho, eps;
};

// Lasagne 0.1 adadelta [upstream]:
//   a = rho*a + (1-rho)*g^2 ; u = g*sqrt(d+eps)/sqrt(a+eps) ; p = p - lr*u ;
//   d = rho*d + (1-rho)*u^2
__device__ __forceinline__ void adadelta_elem(float& p, float& g, float& accu, float& delta,
                                              const AdadeltaArgs& a, float omr, float& ss) {
#pragma clang fp contract(off)
    const float pv = p;
    const float l2 = a.l2k * pv;
    const float gv = g + l2;
    ss = sq_acc(ss, pv);
    const float a1 = a.rho * accu, a2 = (omr * gv) * gv;
    const float av = a1 + a2;
    const float dv = delta;
#ifndef SERT_OPT_IEEE_DIV
    const float s_d = __builtin_amdgcn_sqrtf(dv + a.eps), r_a = __builtin_amdgcn_rsqf(av + a.eps);
    const float u = (gv * s_d) * r_a;
#else
    const float s_d = sqrtf(dv + a.eps), s_a = sqrtf(av + a.eps);
    const float u = (gv * s_d) / s_a;
#endif
    accu = av;
    const float step = a.lr * u;
    p = pv - step;
    const float d1 = a.rho * dv, d2 = (omr * u) * u;
    delta = d1 + d2;
    g = gv;
}

template <bool STORE_G>
__global__ __launch_bounds__(256) void adadelta_l2(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ accu,
                                                   float* __restrict__ delta, size_t count,
                                                   AdadeltaArgs a,
                                                   float* __restrict__ sumsq_partial,
                                                   const uint32_t* __restrict__ bits = nullptr,
                                                   unsigned row_len = 1, int rows_mode = kRowsAll) {
    __shared__ float red[4];
    float ss = 0.f;
    const float omr = 1.0f - a.rho;
    const size_t n4 = count >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* a4 = reinterpret_cast<float4*>(accu);
    float4* d4 = reinterpret_cast<float4*>(delta);
    // each workgroup streams ONE contiguous slice (a grid-stride walk at a 4 MB
    // power-of-two stride keeps all workgroups on the same HBM channels at once)
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const bool hit = !bits || row_bit(bits, ((unsigned)i << 2) / row_len);
        if (rows_mode != kRowsAll && hit != (rows_mode == kRowsTouched)) continue;
        float4 pp = p4[i], aa = a4[i], dd = d4[i];
        float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hit) gg = g4[i];
        adadelta_elem(pp.x, gg.x, aa.x, dd.x, a, omr, ss);
        adadelta_elem(pp.y, gg.y, aa.y, dd.y, a, omr, ss);
        adadelta_elem(pp.z, gg.z, aa.z, dd.z, a, omr, ss);
        adadelta_elem(pp.w, gg.w, aa.w, dd.w, a, omr, ss);
        p4[i] = pp; a4[i] = aa; d4[i] = dd;
        if (STORE_G) g4[i] = gg;
    }
    if (blockIdx.x == 0 && !bits) {
        const size_t i = (n4 << 2) + threadIdx.x;
        if (i < count) {
            float pp = p[i], gg = g[i], aa = accu[i], dd = delta[i];
            adadelta_elem(pp, gg, aa, dd, a, omr, ss);
            p[i] = pp; accu[i] = aa; delta[i] = dd;
            if (STORE_G) g[i] = gg;
        }
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
}

// ---- the word table's dense update, LAZILY but bit for bit -------------------------------------------------------
// The reference moves EVERY row EVERY step (dense L2, sert/models.py:764-795; dense Adam / Adadelta, :548-549) and
// returns a loss that contains sum(p^2): at C4 that is 0.75 ms of a 1.95 ms step streaming p, m, v of 150 M parameters
// in AND out for a batch that touches 14 % of the rows.  But the recurrence of an element no token points to
// (g = lambda/B p) is element-local: a row may stay in memory at the state of update s < t and be brought forward in
// REGISTERS -- t - s applications of the very same adam_elem / adadelta_elem, in the same order on the same inputs, hence
// the same bits -- whenever its current value is needed:
//   * every step for sum(p^2): read p, m, v (12 B per element instead of 24 B read + written);
//   * written back (and advanced by this step's update) only if the batch touches the row (a real gradient), if the
//     NEXT batch will (its forward reads the row: sert_hint_next_batch; no hint = write everything), or every
//     kLazyK-th update (bounds the catch-up loop; 4: with 8 the loop's VALU time eats the saved bytes).
// last[row] = number of updates applied to the stored (p, m, v) of the row; double-buffered (last_in / last_out) because
// the float4 pieces of one row may be walked by two workgroups.  LazyArgs::update = 0 is the FLUSH: every row is brought
// to t_prev and written, nothing else (in front of evaluations, predictions, tensor reads and writes).
#ifndef SERT_LAZY_K
#define SERT_LAZY_K 4
#endif
constexpr int kLazyK = SERT_LAZY_K;
struct LazyArgs {
    const int32_t* last_in;      // null: every row is at t_prev
    int32_t* last_out;
    const uint32_t* next_bits;   // rows the next batch touches, or null
    int t_prev;                  // updates a CURRENT row has seen before this launch
    int write_all;               // materialise every row
    int update;                  // 1: training step (update t_prev + 1), 0: flush
    float a_of[kLazyK + 1];      // Adam: a_of[k] = step size a_t of update number t_prev + 1 - k
};
__device__ __forceinline__ float lazy_step_size(const LazyArgs& lz, int k) {
    float r = lz.a_of[1];
#pragma unroll
    for (int j = 2; j <= kLazyK; ++j) r = (k == j) ? lz.a_of[j] : r;
    return r;
}

template <bool ADAM>
__global__ __launch_bounds__(256) void dense_update_lazy(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ s0, float* __restrict__ s1, size_t count,
                                                         AdamArgs a, AdadeltaArgs da, float* __restrict__ sumsq_partial,
                                                         const uint32_t* __restrict__ bits, unsigned row_len, const LazyArgs lz) {
    __shared__ float red[4];
    float ss = 0.f;
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2, omr = 1.0f - da.rho;
    const size_t n4 = count >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    // (the same contiguous slices and the same thread -> element map as adam_l2 / adadelta_l2: the partial sums of
    //  squares are the same numbers)
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const unsigned e0 = (unsigned)i << 2;
        const unsigned row = e0 / row_len;
        const bool first_piece = e0 - row * row_len == 0u;
        const int last = lz.last_in ? lz.last_in[row] : lz.t_prev;
        const int lag = lz.t_prev - last;
        const bool hit = lz.update && row_bit(bits, row);
        const bool need = lz.update ? (hit || lz.write_all || (lz.next_bits && row_bit(lz.next_bits, row))) : (lag > 0);
        if (first_piece) lz.last_out[row] = lz.update ? (need ? lz.t_prev + 1 : last) : lz.t_prev;
        if (!lz.update && lag == 0) continue;
        float4 pp = p4[i];
        const nt_f4 mr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(s0) + i);
        const nt_f4 vr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(s1) + i);
        float4 mm = make_float4(mr.x, mr.y, mr.z, mr.w), vv = make_float4(vr.x, vr.y, vr.z, vr.w);
        // catch-up: the updates last + 1 .. t_prev, each with a zero gradient (the L2 term alone), as the dense launch of
        // that step applied them to every other row
        for (int k = lag; k >= 1; --k) {
            float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f, sd = 0.f, sn = 0.f;
            if (ADAM) {
                AdamArgs ak = a;
                ak.a_t = lazy_step_size(lz, k);
                adam_elem(pp.x, z0, mm.x, vv.x, ak, omb1, omb2, sd, sn);
                adam_elem(pp.y, z1, mm.y, vv.y, ak, omb1, omb2, sd, sn);
                adam_elem(pp.z, z2, mm.z, vv.z, ak, omb1, omb2, sd, sn);
                adam_elem(pp.w, z3, mm.w, vv.w, ak, omb1, omb2, sd, sn);
            } else {
                adadelta_elem(pp.x, z0, mm.x, vv.x, da, omr, sd);
                adadelta_elem(pp.y, z1, mm.y, vv.y, da, omr, sd);
                adadelta_elem(pp.z, z2, mm.z, vv.z, da, omr, sd);
                adadelta_elem(pp.w, z3, mm.w, vv.w, da, omr, sd);
            }
        }
        if (lz.update && !need) {
            // not written: only its share of sum(p^2) -- the expression and the order of adam_elem / adadelta_elem
            ss = sq_acc(ss, pp.x);
            ss = sq_acc(ss, pp.y);
            ss = sq_acc(ss, pp.z);
            ss = sq_acc(ss, pp.w);
            continue;
        }
        if (lz.update) {
            float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hit) gg = g4[i];
            float sn = 0.f;
            if (ADAM) {
                AdamArgs ak = a;
                ak.a_t = lz.a_of[0];
                adam_elem(pp.x, gg.x, mm.x, vv.x, ak, omb1, omb2, ss, sn);
                adam_elem(pp.y, gg.y, mm.y, vv.y, ak, omb1, omb2, ss, sn);
                adam_elem(pp.z, gg.z, mm.z, vv.z, ak, omb1, omb2, ss, sn);
                adam_elem(pp.w, gg.w, mm.w, vv.w, ak, omb1, omb2, ss, sn);
            } else {
                adadelta_elem(pp.x, gg.x, mm.x, vv.x, da, omr, ss);
                adadelta_elem(pp.y, gg.y, mm.y, vv.y, da, omr, ss);
                adadelta_elem(pp.z, gg.z, mm.z, vv.z, da, omr, ss);
                adadelta_elem(pp.w, gg.w, mm.w, vv.w, da, omr, ss);
            }
        }
        p4[i] = pp;
        { nt_f4 t; t.x = mm.x; t.y = mm.y; t.z = mm.z; t.w = mm.w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(s0) + i); }
        { nt_f4 t; t.x = vv.x; t.y = vv.y; t.z = vv.z; t.w = vv.w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(s1) + i); }
    }
    if (lz.update) {
        const float tot = block_sum_256(ss, red);
        if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
    }
}

// ---- ... and without READING the rows nobody needs ------------------------------------------------------------------
// dense_update_lazy still reads p, m, v of every row every step: sum(p^2) of the loss needs the current value of each
// (12 B per element: the product-search settings' 75-85 us, W3C's 65, C4's 530).  But a row nobody touches follows its
// zero-gradient trajectory, and that trajectory is known when the row is WRITTEN: the launch that writes a row at state
// u also leaves the row's share of sum(p^2) for the updates u + 1 .. T - 1 (T = the next update that reads everything,
// at most kLazyK ahead) in pred[update % kLazyK][row] -- the row sum of p^2 along up to kLazyK - 2 further zero-gradient
// updates in registers, the same adam_elem / adadelta_elem calls on the same inputs the catch-up loop will make later,
// hence the same bits.  A later launch that does not need the row adds pred[...] instead of reading 12 d bytes.
//   * One lane group (32 or 64 lanes) per row; a lane owns float4 columns l, l + LPR, ...; the row's sum of squares is the
//     lanes' fma chains (x, y, z, w, column by column) reduced in a fixed lane order, and a workgroup's partial is the sum
//     of its groups' row sums in row order.  A predicted row sum is bit for bit the one a launch that read the row would
//     have formed, so the partials -- and the loss -- do not depend on which rows were skipped, on the hints or on where
//     the full passes fall.  (They differ from adam_l2's element-order partials in the last bits: a different summation
//     tree over the same squares.  Parameters and optimiser state are the same bits as the dense launches'.)
//   * lz.write_all (full pass): every row is read, brought forward and written -- no prediction is consulted.
//   * The flush (LazyArgs::update = 0) stays with dense_update_lazy: predictions are indexed by update number and a
//     flush moves rows ALONG their trajectories, so they stay valid behind it.
//   * SERT_SKIP_PF (default 1): the pieces of the group's NEXT row are requested before the arithmetic of the current one
//     (a row is three or four 16-byte loads per lane, then up to kLazyK + 2 element updates with a square root and a
//     reciprocal each and two lane-group reductions: without the prefetch a lane group has nothing in flight while it
//     computes).  Same loads, same arithmetic, same bits.  Adam only: measured (tools/experiments/r05_skip_pf.sh, us per launch
//     without / with) C2 51.6 -> 48.1, C4 389 -> 365, product-search settings 66.2 -> 68.0 (the step 0.177 -> 0.175 ms);
//     Adadelta's element update (two square roots, two reciprocals) is VALU-bound and the second set of registers costs
//     it a resident wave: W3C loglinear settings 59.8 -> 65.8 us, so it keeps the plain loop.
#ifndef SERT_SKIP_PF
#define SERT_SKIP_PF 1
#endif
constexpr bool kSkipPrefetchAdam = SERT_SKIP_PF != 0;
//   * SERT_SKIP_WAVES(CPL): minimum waves per SIMD asked of the compiler.  The fixed grid of 2048 workgroups is exactly eight waves per
//     SIMD of 256 CUs and the prefetching kernel takes 72 registers (seven waves fit): asked to stay within 64 (it then spills six
//     dwords) it is SLOWER -- C2 0.2394 -> 0.2439 ms, tools/experiments/r05_skip_waves.sh.  Not asked.
#ifndef SERT_SKIP_WAVES
#define SERT_SKIP_WAVES(CPL) 1
#endif
struct SkipArgs {
    float* pred;             // [kLazyK][stride]
    unsigned stride;
    int npred;               // a written row leaves predictions for the updates t_prev + 2 .. t_prev + 1 + npred
    float a_fut[kLazyK];     // Adam: step size of update t_prev + 2 + j
};

template <int LPR>
__device__ __forceinline__ float lane_group_sum(float v, int lane) {
    v = row16_sum(v);
    const float lo = read_lane(v, 0) + read_lane(v, 16), hi = read_lane(v, 32) + read_lane(v, 48);
    if (LPR == 64) return lo + hi;
    return lane < 32 ? lo : hi;
}

template <bool ADAM, int LPR, int CPL>
__global__ __launch_bounds__(256, SERT_SKIP_WAVES(CPL)) void dense_update_skip(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ s0, float* __restrict__ s1, unsigned nrows,
                                                         AdamArgs a, AdadeltaArgs da, float* __restrict__ sumsq_partial,
                                                         const uint32_t* __restrict__ bits, unsigned row_len, const LazyArgs lz,
                                                         const SkipArgs sk) {
    constexpr int GPB = 256 / LPR;      // lane groups = rows in flight per workgroup
    constexpr bool kSkipPrefetch = kSkipPrefetchAdam && ADAM;
    __shared__ float red[GPB];
    const int lane = threadIdx.x & 63, l = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2, omr = 1.0f - da.rho;
    const unsigned d4 = row_len >> 2;
    const unsigned per = (nrows + gridDim.x - 1) / gridDim.x;
    const unsigned lo = blockIdx.x * per;
    const unsigned hi = lo + per < nrows ? lo + per : nrows;
    const int slot_now = (lz.t_prev + 1) % kLazyK;
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float acc = 0.f;                    // this group's row sums, added in row order (the same value in every lane of the group)
    // rows of this group: lo + grp, lo + grp + GPB, ...; the first group of a wave has the most
    const unsigned cnt = hi > lo + grp ? (hi - lo - grp + GPB - 1) / GPB : 0u;
    const unsigned cnt_w = (unsigned)__builtin_amdgcn_readfirstlane((int)cnt);
    for (unsigned j0 = 0; j0 < cnt_w; j0 += LPR) {
        // what to do with the next LPR rows of the group: one row per lane (bitmaps, update counter, prediction)
        const unsigned jm = j0 + l;
        const unsigned row_m = lo + grp + jm * GPB;
        int flags_m = 0, last_m = lz.t_prev;
        float pred_m = 0.f;
        if (jm < cnt) {
            last_m = lz.last_in ? lz.last_in[row_m] : lz.t_prev;
            const bool hit_m = row_bit(bits, row_m);
            const bool need_m = hit_m || lz.write_all || (lz.next_bits && row_bit(lz.next_bits, row_m));
            lz.last_out[row_m] = need_m ? lz.t_prev + 1 : last_m;
            if (!need_m) pred_m = sk.pred[(size_t)slot_now * sk.stride + row_m];
            flags_m = 1 | (hit_m ? 2 : 0) | (need_m ? 4 : 0);
        }
        const unsigned trips = cnt_w - j0 < (unsigned)LPR ? cnt_w - j0 : (unsigned)LPR;
        // a row's pieces: fetched one trip AHEAD of the arithmetic (SERT_SKIP_PF, see above the kernel)
        struct RowRegs { float4 p[CPL], m[CPL], v[CPL], g[CPL]; };
        auto fetch = [&](unsigned jj, RowRegs& r) {
            const int flags = __shfl(flags_m, (int)jj, LPR);
            const bool hit = flags & 2, proc = (flags & 5) == 5;
            const unsigned row = lo + grp + (j0 + jj) * GPB;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                r.p[c] = r.m[c] = r.v[c] = r.g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (proc && (unsigned)(l + c * LPR) < d4) {
                    const size_t i = (size_t)row * d4 + l + c * LPR;
                    r.p[c] = p4[i];
                    const nt_f4 mr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(s0) + i);
                    const nt_f4 vr = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(s1) + i);
                    r.m[c] = make_float4(mr.x, mr.y, mr.z, mr.w);
                    r.v[c] = make_float4(vr.x, vr.y, vr.z, vr.w);
                    if (hit) r.g[c] = g4[i];
                }
            }
        };
        RowRegs nxt;
        if (kSkipPrefetch && trips > 0) fetch(0, nxt);
        for (unsigned jj = 0; jj < trips; ++jj) {
            const int flags = __shfl(flags_m, (int)jj, LPR);
            const int last = __shfl(last_m, (int)jj, LPR);
            const float predv = __shfl(pred_m, (int)jj, LPR);
            const bool valid = flags & 1, proc = (flags & 5) == 5;
            RowRegs cur;
            if (kSkipPrefetch) {
                cur = nxt;
                if (jj + 1 < trips) fetch(jj + 1, nxt);
            }
            if (!__any(proc)) {          // (wave-uniform) nothing to read: the predicted shares
                if (valid) acc += predv;
                continue;
            }
            if (!kSkipPrefetch) fetch(jj, cur);
            const unsigned row = lo + grp + (j0 + jj) * GPB;
            const int lag = proc ? lz.t_prev - last : 0;
            bool on[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) on[c] = proc && (unsigned)(l + c * LPR) < d4;
            float4 (&pp)[CPL] = cur.p;
            float4 (&mm)[CPL] = cur.m;
            float4 (&vv)[CPL] = cur.v;
            // one zero-gradient update of the lane's columns with Adam step size at; s collects the squares BEFORE it
            auto advance = [&](float at, float& s) {
                float sn = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (!on[c]) continue;
                    float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
                    if (ADAM) {
                        AdamArgs ak = a;
                        ak.a_t = at;
                        adam_elem(pp[c].x, z0, mm[c].x, vv[c].x, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].y, z1, mm[c].y, vv[c].y, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].z, z2, mm[c].z, vv[c].z, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].w, z3, mm[c].w, vv[c].w, ak, omb1, omb2, s, sn);
                    } else {
                        adadelta_elem(pp[c].x, z0, mm[c].x, vv[c].x, da, omr, s);
                        adadelta_elem(pp[c].y, z1, mm[c].y, vv[c].y, da, omr, s);
                        adadelta_elem(pp[c].z, z2, mm[c].z, vv[c].z, da, omr, s);
                        adadelta_elem(pp[c].w, z3, mm[c].w, vv[c].w, da, omr, s);
                    }
                }
            };
            // catch-up: the updates last + 1 .. t_prev the row sat out
            for (int k = kLazyK; k >= 1; --k) {
                if (k <= lag) { float sd = 0.f; advance(lazy_step_size(lz, k), sd); }
            }
            // this step's update
            float s = 0.f;
            {
                float sn = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (!on[c]) continue;
                    const size_t i = (size_t)row * d4 + l + c * LPR;
                    float4 gg = cur.g[c];       // (zero unless a token of the batch points to the row)
                    if (ADAM) {
                        AdamArgs ak = a;
                        ak.a_t = lz.a_of[0];
                        adam_elem(pp[c].x, gg.x, mm[c].x, vv[c].x, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].y, gg.y, mm[c].y, vv[c].y, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].z, gg.z, mm[c].z, vv[c].z, ak, omb1, omb2, s, sn);
                        adam_elem(pp[c].w, gg.w, mm[c].w, vv[c].w, ak, omb1, omb2, s, sn);
                    } else {
                        adadelta_elem(pp[c].x, gg.x, mm[c].x, vv[c].x, da, omr, s);
                        adadelta_elem(pp[c].y, gg.y, mm[c].y, vv[c].y, da, omr, s);
                        adadelta_elem(pp[c].z, gg.z, mm[c].z, vv[c].z, da, omr, s);
                        adadelta_elem(pp[c].w, gg.w, mm[c].w, vv[c].w, da, omr, s);
                    }
                    p4[i] = pp[c];
                    { nt_f4 t; t.x = mm[c].x; t.y = mm[c].y; t.z = mm[c].z; t.w = mm[c].w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(s0) + i); }
                    { nt_f4 t; t.x = vv[c].x; t.y = vv[c].y; t.z = vv[c].z; t.w = vv[c].w; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(s1) + i); }
                }
            }
            const float rowsum = lane_group_sum<LPR>(s, lane);
            if (valid) acc += proc ? rowsum : predv;
            // the written row's shares of sum(p^2) while nobody reads it
            for (int q = 1; q <= sk.npred; ++q) {
                float s2 = 0.f;
                if (q < sk.npred) {
                    advance(sk.a_fut[q - 1], s2);
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        if (!on[c]) continue;
                        s2 = sq_acc(s2, pp[c].x); s2 = sq_acc(s2, pp[c].y); s2 = sq_acc(s2, pp[c].z); s2 = sq_acc(s2, pp[c].w);
                    }
                }
                const float r = lane_group_sum<LPR>(s2, lane);
                if (proc && l == 0) sk.pred[(size_t)((lz.t_prev + 1 + q) % kLazyK) * sk.stride + row] = r;
            }
        }
    }
    if (l == 0) red[grp] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int i = 1; i < GPB; ++i) t += red[i];
        sumsq_partial[blockIdx.x] = t;
    }
}

// The small tensors (entity table at small V_e, dense W, bias) in ONE launch: a
// kernel boundary costs more than updating them.  Block b works on the tensor
// whose block range contains it; partial sums of squares of non-regularised
// tensors (the bias) are written as 0.
struct SmallTensors {
    float* p[3];
    float* g[3];
    float* s0[3];
    float* s1[3];
    unsigned long long count[3];
    int first_block[4];     // block range of tensor i = [first_block[i], first_block[i+1])
    float l2k[3];           // lambda/B, or 0 for a tensor without L2
    // gradient still in `ngroups` partial tables `gstride` elements apart (the entity table behind
    // egrad_acc): added here in group order -- the sum egrad_group_sum would have made in a launch
    // of its own -- and stored to g as well
    const float* gparts[3];
    int ngroups[3];
    unsigned long long gstride[3];
};

// sumsq_new_partial (optional): also the sums of squares of the UPDATED values, partitioned exactly as sumsq_partial
// -- bit for bit what the next step's launch over the same tensors computes as its pre-update sums (the small
// entity table's update deferred past the tail, sert_hip.hip: defer_small).
template <bool ADAM, bool STORE_G>
__global__ __launch_bounds__(256) void optimizer_small(SmallTensors t, AdamArgs aa, AdadeltaArgs da,
                                                       float* __restrict__ sumsq_partial,
                                                       float* __restrict__ sumsq_new_partial = nullptr) {
    __shared__ float red[4];
    int i = 0;
    if ((int)blockIdx.x >= t.first_block[1]) i = 1;
    if ((int)blockIdx.x >= t.first_block[2]) i = 2;
    const int nb = t.first_block[i + 1] - t.first_block[i];
    const int b = blockIdx.x - t.first_block[i];
    float *p = t.p[i], *g = t.g[i], *s0 = t.s0[i], *s1 = t.s1[i];
    const size_t count = (size_t)t.count[i];
    aa.l2k = t.l2k[i];
    da.l2k = t.l2k[i];
    const float omb1 = 1.0f - aa.b1, omb2 = 1.0f - aa.b2, omr = 1.0f - da.rho;
    float ss = 0.f, ssn = 0.f;
    const float* parts = t.gparts[i];
    const int ngroups = t.ngroups[i];
    const size_t gstride = (size_t)t.gstride[i];
    for (size_t k = (size_t)b * 256 + threadIdx.x; k < count; k += (size_t)nb * 256) {
        float pp = p[k], a0 = s0[k], a1 = s1[k];
        float gg;
        if (parts) {
            gg = parts[k];
            for (int q = 1; q < ngroups; ++q) gg += parts[(size_t)q * gstride + k];
            if (!STORE_G) g[k] = gg;
        } else {
            gg = g[k];
        }
        if (ADAM) adam_elem(pp, gg, a0, a1, aa, omb1, omb2, ss, ssn);
        else { adadelta_elem(pp, gg, a0, a1, da, omr, ss); ssn = sq_acc(ssn, pp); }
        p[k] = pp; s0[k] = a0; s1[k] = a1;
        if (STORE_G) g[k] = gg;
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = (t.l2k[i] != 0.f) ? tot : 0.f;
    if (sumsq_new_partial) {      // (kernel-uniform)
        __syncthreads();
        const float tn = block_sum_256(ssn, red);
        if (threadIdx.x == 0) sumsq_new_partial[blockIdx.x] = (t.l2k[i] != 0.f) ? tn : 0.f;
    }
}

// The pre-update sums of squares of optimizer_small over ONE tensor WITHOUT the update: same strided shares, same
// element order, same block reduction -- partial[b] equals what optimizer_small leaves in sumsq_partial[first_block + b]
// (first step of a deferred small update, or after the host replaced the table).
__global__ __launch_bounds__(256) void sumsq_like_small(const float* __restrict__ p, size_t count, float l2k,
                                                        float* __restrict__ partial) {
    __shared__ float red[4];
    float ss = 0.f;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < count; k += (size_t)gridDim.x * 256) ss = sq_acc(ss, p[k]);
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = (l2k != 0.f) ? tot : 0.f;
}

// partial[b] = sum of block b's strided share of in[0..count)
__global__ __launch_bounds__(256) void sum_partial(const float* __restrict__ in, size_t count,
                                                   float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        s += in[i];
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Sum of squares over the `nch` pieces of `sc` elements (sc % 4 == 0) that start at p and lie
// `slab` elements apart (the pieces of a ZeRO-sharded tensor one rank owns): per-block
// partials, fixed grid => fixed tree.
__global__ __launch_bounds__(256) void sumsq_pieces(const float* __restrict__ p, size_t sc, size_t slab,
                                                    int nch, float* __restrict__ partial) {
    __shared__ float red[4];
    float ss = 0.f;
    const size_t sc4 = sc >> 2, total4 = sc4 * (size_t)nch;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total4; j += (size_t)gridDim.x * blockDim.x) {
        const size_t c = j / sc4, o = j - c * sc4;
        const float4 v = reinterpret_cast<const float4*>(p + c * slab)[o];
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// out[0] = sum of n partials (fp64, fixed order): the local loss sum that travels in
// the flat gradient buffer of a data-parallel step
__global__ __launch_bounds__(256) void partials_to_scalar(const float* __restrict__ partials, int n,
                                                          float* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)partials[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// loss = (sum rowloss)/B + lambda/(2B) * sum of squares of the regularised
// tensors (models.py:278-282, :773-791).  Single block, fp64 accumulation of
// the (few thousand) partials, fixed order.
//   parts: [loss partials (n_loss)] [sumsq partials (n_sq)]
__global__ __launch_bounds__(256) void finalize_loss(const float* __restrict__ loss_partials,
                                                     int n_loss,
                                                     const float* __restrict__ sq_partials,
                                                     int n_sq, float inv_batch, float reg_scale,
                                                     float* __restrict__ out,
                                                     unsigned* __restrict__ host_flag = nullptr,
                                                     unsigned seq = 0,
                                                     const float* __restrict__ extra_sq = nullptr) {
    // eight partials per thread and trip in flight (the loop used to be one dependent load per
    // partial: 16 + 9 round trips in series at C2), both sums reduced together: lanes by shuffle,
    // the four waves through LDS -- one barrier instead of eighteen.  Fixed order throughout.
    __shared__ double red[2][4];
    auto strided_sum = [](const float* __restrict__ x, int n) {
        double acc = 0.0;
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (i0 + q * 256 < n) ? x[i0 + q * 256] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (double)v[q];
        }
        return acc;
    };
    double a = strided_sum(loss_partials, n_loss);
    double b = strided_sum(sq_partials, n_sq);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    const double loss_sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double sq_sum = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (threadIdx.x == 0) {
        const float data = (float)loss_sum * inv_batch;
        // (data parallel: + the all-reduced sum of squares of the ZeRO-sharded tensors)
        const float reg = reg_scale * (float)(sq_sum + (extra_sq ? (double)extra_sq[0] : 0.0));
        out[0] = data + reg;
        out[1] = data;
        out[2] = reg;
        // out may be pinned host memory: publish the step's sequence number after the
        // values (system-scope release) -- the host spins on it instead of a stream sync
        if (host_flag) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the tail of a single-GPU vectorspace step in ONE launch ---------------------------------------
// split-K combine of dW / db (gemm.h: reduce_partials_g<16>, same association) -> Adam on W and b ->
// loss finalisation (finalize_loss above).  Three dependent launches of 4-13 us each were ~28 us of a
// 330 us step, most of it launch and drain latency.  Every workgroup also pre-reduces a slice of the
// loss and sum-of-squares partials (fp64) and publishes its two sums; the workgroup with the HIGHEST
// index (dispatched last) collects them in workgroup order and publishes the loss -- a fixed
// association.  No fence anywhere: a published value is ONE 64-bit agent-scope store that carries the
// launch's sequence number in its upper half, so value and "ready" flag arrive together (a release
// fence per workgroup -- an L2 write-back on this multi-XCD part -- made the first version of this
// kernel take 23 us; nothing but these words has to be visible inside the launch).  The collector
// never blocks anyone: every other workgroup runs to completion without waiting.
struct TailArgs {
    const float* part;            // [splits][stride]: dW (n_w) then db (n_b) partial slabs
    int splits;
    unsigned long long stride;
    float *W, *b, *s0_w, *s1_w, *s0_b, *s1_b, *g_w, *g_b;
    unsigned n_w, n_b;
    AdamArgs aa;                  // l2k applies to W only (the bias is not regularised)
    const float* loss_partials; int n_loss;
    const float* sq_partials; int n_sq;    // sums of squares of the tensors updated before this launch
    const float* sq_alt; int sq_alt_lo, sq_alt_hi;   // partials [sq_alt_lo, sq_alt_hi) live in sq_alt[0..) instead (empty range: none)
    float inv_batch, reg_scale;
    float* out;                   // [3] loss, data term, reg term (device or pinned host)
    unsigned* host_flag; unsigned seq;
    unsigned long long* blk;      // [4 * gridDim.x] (launch sequence number << 32) | float bits: loss hi, lo, squares hi, lo
    unsigned launch_seq;          // != 0, different from the previous launch's
};

template <bool STORE_G>
__global__ __launch_bounds__(1024) void vs_tail(const TailArgs t) {
    __shared__ float red[16][64];
    __shared__ double dred[2][16];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const unsigned count = t.n_w + t.n_b;
    const unsigned i = blockIdx.x * 64u + (unsigned)l;
    float a = 0.f;
    if (i < count) {
#pragma unroll 8
        for (int s = g; s < t.splits; s += 16) a += t.part[(size_t)s * t.stride + i];
    }
    red[g][l] = a;
    double ls = 0.0, sq = 0.0;
    for (int k = blockIdx.x * 1024 + threadIdx.x; k < t.n_loss; k += gridDim.x * 1024) ls += (double)t.loss_partials[k];
    for (int k = blockIdx.x * 1024 + threadIdx.x; k < t.n_sq; k += gridDim.x * 1024)
        sq += (double)((k >= t.sq_alt_lo && k < t.sq_alt_hi) ? t.sq_alt[k - t.sq_alt_lo] : t.sq_partials[k]);
    __syncthreads();
    if (g == 0 && i < count) {
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (red[4 * k][l] + red[4 * k + 1][l]) + (red[4 * k + 2][l] + red[4 * k + 3][l]);
        float gg = ((q[0] + q[1]) + q[2]) + q[3];   // (= reduce_partials_g<16>)
        AdamArgs aa = t.aa;
        const float omb1 = 1.0f - aa.b1, omb2 = 1.0f - aa.b2;
        float ssf = 0.f;
        if (i < t.n_w) {
            float pp = t.W[i], m = t.s0_w[i], v = t.s1_w[i];
            { float unused = 0.f; adam_elem(pp, gg, m, v, aa, omb1, omb2, ssf, unused); }
            t.W[i] = pp; t.s0_w[i] = m; t.s1_w[i] = v;
            if (STORE_G) t.g_w[i] = gg;
            sq += (double)ssf;
        } else {
            const unsigned j = i - t.n_w;
            aa.l2k = 0.f;
            float pp = t.b[j], m = t.s0_b[j], v = t.s1_b[j];
            { float unused = 0.f; adam_elem(pp, gg, m, v, aa, omb1, omb2, ssf, unused); }
            t.b[j] = pp; t.s0_b[j] = m; t.s1_b[j] = v;
            if (STORE_G) t.g_b[j] = gg;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ls += __shfl_xor(ls, off);
        sq += __shfl_xor(sq, off);
    }
    if (l == 0) { dred[0][g] = ls; dred[1][g] = sq; }
    __syncthreads();
    if (threadIdx.x < 4) {
        // each fp64 sum travels as a (hi, lo) pair of floats -- hi = (float)x, lo = (float)(x - hi): 48 bits of the
        // double survive, so the loss this launch returns equals finalize_loss's (fp64 to the end, as the
        // reference's float64 Sum) far below one float32 ulp instead of carrying ~1000 extra roundings
        const int which = threadIdx.x >> 1;
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) x += dred[which][w];
        const float hi = (float)x;
        const float piece = (threadIdx.x & 1) ? (float)(x - (double)hi) : hi;
        const unsigned long long word = ((unsigned long long)t.launch_seq << 32) | (unsigned long long)__float_as_uint(piece);
        __hip_atomic_store(t.blk + 4 * blockIdx.x + threadIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x != gridDim.x - 1) return;
    // ---- the collector: wait for every workgroup's four words, add them in workgroup order ----
    // (the only wait of the launch; it cannot deadlock: no other workgroup ever waits, so they all run to
    // completion whatever the dispatch order -- also under serialised dispatch, where this one is simply last)
    __syncthreads();
    double c0 = 0.0, c1 = 0.0;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += 1024) {
        unsigned long long w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            do { w[j] = __hip_atomic_load(t.blk + 4 * k + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(w[j] >> 32) != t.launch_seq);
        c0 += (double)__uint_as_float((unsigned)w[0]) + (double)__uint_as_float((unsigned)w[1]);
        c1 += (double)__uint_as_float((unsigned)w[2]) + (double)__uint_as_float((unsigned)w[3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_xor(c0, off);
        c1 += __shfl_xor(c1, off);
    }
    if (l == 0) { dred[0][g] = c0; dred[1][g] = c1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double loss_sum = 0.0, sq_sum = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { loss_sum += dred[0][w]; sq_sum += dred[1][w]; }
        const float data = (float)loss_sum * t.inv_batch;
        const float reg = t.reg_scale * (float)sq_sum;
        t.out[0] = data + reg;
        t.out[1] = data;
        t.out[2] = reg;
        if (t.host_flag) __hip_atomic_store(t.host_flag, t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace sert
