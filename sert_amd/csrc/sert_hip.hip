// libsert_hip.so -- C ABI (include/sert_hip.h) over the gfx950 kernels.
// Host orchestration of one training / evaluation / prediction step: the body of
// the reference's compiled Theano functions (sert/models.py:581-608).
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

#include "common.h"
#include "gemm.h"
#include "gemm_x3.h"
#ifdef SERT_VARIANTS   // opt-in variants that lost their A/B (csrc/variants/; tools/build_variant.sh -DSERT_VARIANTS)
#include "variants/gemm_bwd_fused.h"
#include "variants/gemm_big.h"
#include "variants/gemm_x3_bres.h"
#include "variants/score_filter_ring.h"
#include "variants/gemm_strip.h"
#include "variants/gemm_stream.h"
#include "variants/gemm_direct.h"
#endif
#include "kernels_score_bf16.h"
#include "kernels_egrad.h"
#include "kernels_ll.h"
#include "kernels_membench.h"
#include "kernels_opt.h"
#include "kernels_score.h"
#include "kernels_seg.h"
#include "kernels_sort.h"
#include "kernels_vs.h"
#ifdef SERT_VARIANTS
#include "variants/kernels_gather_hot.h"
#include "variants/kernels_seg_bundled.h"
#include "variants/kernels_egrad_ranges.h"
#include "variants/kernels_proj.h"   // (after kernels_vs.h / gemm_x3.h: gather + mean-pool + projection in one persistent launch, measured slower)
#endif
#include "model.h"

namespace sert {
thread_local std::string g_last_error;

static const char* kTimingNames[TG_COUNT] = {
    "gather",        "gemm_fwd", "loss",      "entity_sort",          "entity_grad_reduce",
    "entity_grad_fixup", "gemm_dW", "splitk_combine", "gemm_dX",      "word_grad_segsum",
    "allreduce",     "reduce_scatter", "all_gather", "optimizer_word_table",  "optimizer_other",      "finalize"};

// ---- roctx ranges (sert_profile_range_push / pop) ------------------------------------
// Loaded lazily from the ROCm tools library; SERT_ROCTX=1 additionally wraps every kernel group of a step
// (the timing groups below) in a range, so that a rocprofv3 --marker-trace shows the step's structure on
// the host timeline.  Without the library the calls are no-ops.
struct Roctx {
    bool tried = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static Roctx g_roctx;
static void roctx_load() {
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    for (const char* n : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/libroctx64.so"}) {
        void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) continue;
        g_roctx.push = (decltype(g_roctx.push))dlsym(lib, "roctxRangePushA");
        g_roctx.pop = (decltype(g_roctx.pop))dlsym(lib, "roctxRangePop");
        if (g_roctx.push && g_roctx.pop) return;
        g_roctx.push = nullptr; g_roctx.pop = nullptr;
    }
}
static bool roctx_groups() {
    static const bool on = [] {
        const bool want = knob("SERT_ROCTX") && atoi(knob("SERT_ROCTX")) != 0;
        if (want) roctx_load();
        return want && g_roctx.push != nullptr;
    }();
    return on;
}

// ---- timing ----------------------------------------------------------------
// in-step mode (sert_timing_enable(m, 2); common.h: InStepHook): a (start, stop) pair for the next launch of the group the
// launching thread is inside
static void instep_harvest(sert_model* m, bool all) {
    InStep& t = m->instep;
    while (t.head < t.tail && (all || t.tail - t.head >= InStep::kRing)) {
        const int i = (int)(t.head % InStep::kRing);
        float ms = 0.f;
        if (hipEventSynchronize(t.ev[i][1]) == hipSuccess && hipEventElapsedTime(&ms, t.ev[i][0], t.ev[i][1]) == hipSuccess) {
            t.total_us[t.group[i]] += 1000.0 * ms;
            t.launches[t.group[i]] += 1;
        }
        ++t.head;
    }
}
static void instep_next(void* ctx, hipEvent_t* a, hipEvent_t* b) {
    sert_model* m = (sert_model*)ctx;
    InStep& t = m->instep;
    if (!t.on || t.cur_group < 0) return;
    instep_harvest(m, false);
    const int i = (int)(t.tail % InStep::kRing);
    t.group[i] = t.cur_group;
    *a = t.ev[i][0];
    *b = t.ev[i][1];
    ++t.tail;
}

struct ScopedTimer {
    sert_model* m;
    int g;
    hipStream_t s;
    int instep_prev = -1;
    InStepHook hook_prev = {nullptr, nullptr};
    ScopedTimer(sert_model* m_, int g_, hipStream_t s_ = nullptr) : m(m_), g(g_), s(s_ ? s_ : m_->stream) {
        if (m->instep.on) {
            instep_prev = m->instep.cur_group;
            hook_prev = instep_hook();
            m->instep.cur_group = g;
            instep_hook() = InStepHook{instep_next, m};
        }
        if (roctx_groups()) (void)g_roctx.push(kTimingNames[g]);
        // a group bracketed several times in one step spans first start .. last end
        if (m->timing.enabled && !m->timing.used[g]) {
            (void)hipEventRecord(m->timing.ev[g][0], s);
        }
    }
    ~ScopedTimer() {
        if (m->timing.enabled) {
            (void)hipEventRecord(m->timing.ev[g][1], s);
            m->timing.used[g] = true;
        }
        if (roctx_groups()) (void)g_roctx.pop();
        if (m->instep.on) {
            m->instep.cur_group = instep_prev;
            instep_hook() = hook_prev;
        }
    }
};

static void timing_collect(sert_model* m) {
    if (!m->timing.enabled) return;
    for (int g = 0; g < TG_COUNT; ++g) {
        if (!m->timing.used[g]) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->timing.ev[g][0], m->timing.ev[g][1]) == hipSuccess) {
            m->timing.total_us[g] += 1000.0 * ms;
            m->timing.samples[g] += 1;
        }
        m->timing.used[g] = false;
    }
}

// ---- RCCL, loaded lazily (single-GPU runs never touch it) --------------------
struct UniqueId {
    char internal[SERT_COMM_ID_BYTES];
};
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /*ncclUniqueId by value*/ UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static Rccl g_rccl;

static int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) SERT_FAIL(std::string("cannot dlopen librccl: ") + dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))dlsym(lib, "ncclReduceScatter");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.Send = (decltype(g_rccl.Send))dlsym(lib, "ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(lib, "ncclRecv");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(lib, "ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce ||
        !g_rccl.ReduceScatter || !g_rccl.AllGather || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart ||
        !g_rccl.GroupEnd)
        SERT_FAIL("librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce/"
                  "ncclReduceScatter/ncclAllGather/ncclSend/ncclRecv/ncclGroupStart/ncclGroupEnd");
    g_rccl.lib = lib;
    return 0;
}
#define SERT_NCCL(expr)                                                                       \
    do {                                                                                      \
        int _r = (expr);                                                                      \
        if (_r != 0)                                                                          \
            SERT_FAIL(std::string(#expr) + ": " +                                             \
                      (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error"));    \
    } while (0)

// ---- small helpers -----------------------------------------------------------
template <typename T>
static int dmalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) return 0;
    SERT_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return 0;
}
template <typename T>
static int dzalloc(T** p, size_t count, hipStream_t s) {
    SERT_TRY(dmalloc(p, count));
    if (count) SERT_HIP(hipMemsetAsync(*p, 0, count * sizeof(T), s));
    return 0;
}
static inline int grid_for(int64_t work_items, int block = 256, int cap = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
static inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// vectorspace-shaped parameters (R_e, projection W/b): both the reference's NCE
// model and the additive full-softmax variant
static bool is_vs(const sert_model* m) { return m->cfg.kind != SERT_KIND_LOGLINEAR; }
static bool is_fs(const sert_model* m) { return m->cfg.kind == SERT_KIND_VECTORSPACE_SOFTMAX; }
// drop whatever a previous training call ran ahead for the next one (sert_hint_next_batch)
static void discard_run_ahead(sert_model* m) {
    if (m->spec_fb_batch >= 0 && m->stream2 && m->ev_join) {
        // a discarded run-ahead may still be busy on the side stream (its entity chain is only
        // joined by the update that will now never come): order the main stream behind it
        // before anything reuses the buffers it reads and writes
        (void)hipEventRecord(m->ev_join, m->stream2);
        (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
    }
    m->spec_fb_batch = -1;
}
static int ensure_rw_current(sert_model* m, int64_t batch, int64_t t_applied = -1);
static bool egrad_writes_every_row(const sert_model* m);
static void invalidate_speculation(sert_model* m) {
    discard_run_ahead(m);
    (void)ensure_rw_current(m, -1);   // (whatever comes next -- new parameters, another step counter, new data -- sees every row current)
    if (m->re_pending) {       // (a deferred entity-table update: see settle_entity_update)
        (void)hipStreamWaitEvent(m->stream, m->ev_re, 0);
        m->re_pending = false;
    }
    m->re_sq_for[0] = m->re_sq_for[1] = -1;
    m->projected_batch = -1;
    m->neg_alt_step = -1;      // (step counter, seed-relevant state or data may change)
    m->rw_pred_ok = false;     // (the rows' predicted shares of sum(p^2) are trajectories of THESE parameters and THIS step counter)
}

// ---- lazy dense update of the word table (kernels_opt.h: dense_update_lazy) ----------------------------------------
static void optimizer_args(const sert_model* m, int64_t t, AdamArgs* aa, AdadeltaArgs* da);
static LazyArgs lazy_args(sert_model* m, int64_t t_prev, int update) {
    LazyArgs lz;
    lz.last_in = m->rw_stale ? m->rw_last[m->rw_last_cur] : nullptr;
    lz.last_out = m->rw_last[m->rw_last_cur ^ 1];
    lz.next_bits = nullptr;
    lz.t_prev = (int)t_prev;
    lz.write_all = 1;
    lz.update = update;
    for (int k = 0; k <= kLazyK; ++k) {
        AdamArgs aa; AdadeltaArgs da;
        optimizer_args(m, std::max<int64_t>(1, t_prev + 1 - k), &aa, &da);
        lz.a_of[k] = aa.a_t;
    }
    return lz;
}
// Bring every row of R_w (and of its optimiser state) to the model's step -- unless `batch` is the training batch
// whose rows the last update made current (the announced one: its forward may read them as they are).  Main stream.
// t_applied = number of updates a CURRENT row has seen (default: m->step).  Inside optimizer_and_loss the step counter is
// already the number of the update being applied, so the dense fallback there passes m->step - 1: flushing against the
// incremented counter gave every row one zero-gradient update too many before the dense launch applied the same update
// number again (round-4 advisor finding; tests/test_gpu_lazy_dense.py alternates lazy and dense steps).
static int ensure_rw_current(sert_model* m, int64_t batch, int64_t t_applied) {
    if (!m->rw_stale) return 0;
    if (batch >= 0 && batch == m->rw_ready_batch) return 0;
    if (t_applied < 0) t_applied = m->step;
    AdamArgs aa; AdadeltaArgs da;
    optimizer_args(m, std::max<int64_t>(1, t_applied), &aa, &da);
    const LazyArgs lz = lazy_args(m, t_applied, /*update=*/0);
    const int64_t max_nb = m->n_rw >= ((size_t)1 << 24) ? 2 * kOptBlocks : kOptBlocks;
    const int nb = (int)std::min<int64_t>(max_nb, cdiv(cdiv(m->n_rw, 4), 256));
    if (is_vs(m))
        hipLaunchKernelGGL((dense_update_lazy<true>), dim3(nb), dim3(256), 0, m->stream, m->rw, (const float*)m->g_rw, m->s0_rw, m->s1_rw,
                           m->n_rw, aa, da, m->sq_scratch, (const uint32_t*)nullptr, (unsigned)m->cfg.word_dim, lz);
    else
        hipLaunchKernelGGL((dense_update_lazy<false>), dim3(nb), dim3(256), 0, m->stream, m->rw, (const float*)m->g_rw, m->s0_rw, m->s1_rw,
                           m->n_rw, aa, da, m->sq_scratch, (const uint32_t*)nullptr, (unsigned)m->cfg.word_dim, lz);
    m->rw_last_cur ^= 1;
    m->rw_stale = false;
    m->rw_ready_batch = -1;
    return 0;
}

// A deferred entity-table update (side-heavy schedule) must have landed before the main stream reads R_e,
// its optimiser state or dR_e again.
static int settle_entity_update(sert_model* m) {
    if (m->re_pending) {
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_re, 0));
        m->re_pending = false;
    }
    return 0;
}

struct TensorRef {
    float* ptr;
    size_t count;
};
static TensorRef tensor_ref(sert_model* m, int which) {
    const size_t B = m->cfg.batch_size;
    switch (which) {
        case SERT_T_RW: return {m->rw, m->n_rw};
        case SERT_T_RE: return {m->re, m->n_re};
        case SERT_T_W: return {m->W, m->n_w};
        case SERT_T_B: return {m->b, m->n_b};
        case SERT_T_STATE0_RW: return {m->s0_rw, m->n_rw};
        case SERT_T_STATE0_RE: return {m->s0_re, m->n_re};
        case SERT_T_STATE0_W: return {m->s0_w, m->n_w};
        case SERT_T_STATE0_B: return {m->s0_b, m->n_b};
        case SERT_T_STATE1_RW: return {m->s1_rw, m->n_rw};
        case SERT_T_STATE1_RE: return {m->s1_re, m->n_re};
        case SERT_T_STATE1_W: return {m->s1_w, m->n_w};
        case SERT_T_STATE1_B: return {m->s1_b, m->n_b};
        case SERT_T_GRAD_RW: return {m->g_rw, m->n_rw};
        case SERT_T_GRAD_RE: return {m->g_re, m->n_re};
        case SERT_T_GRAD_W: return {m->g_w, m->n_w};
        case SERT_T_GRAD_B: return {m->g_b, m->n_b};
        case SERT_T_ACT_H: return {m->H, m->H ? B * m->cfg.word_dim : 0};
        case SERT_T_ACT_T: return {m->T, m->T ? B * m->cfg.entity_dim : 0};
        case SERT_T_ACT_DA: return {m->DA, m->DA ? B * m->cfg.entity_dim : 0};
        case SERT_T_ACT_DH: return {m->DH, m->DH ? B * m->cfg.word_dim : 0};
        case SERT_T_ACT_ROWLOSS: return {m->rowloss, B};
        default: return {nullptr, 0};
    }
}

// Dispatch on the token-id width (np.min_scalar_type, bin/prepare.py:380).
#define SERT_ID_DISPATCH(id_bytes, ...)                                  \
    do {                                                                 \
        if ((id_bytes) == 1) { typedef uint8_t IdT; __VA_ARGS__; }       \
        else if ((id_bytes) == 2) { typedef uint16_t IdT; __VA_ARGS__; } \
        else { typedef uint32_t IdT; __VA_ARGS__; }                      \
    } while (0)


// dR_w = scatter-add of `src` rows into the word table, as an order-fixed
// segmented reduction over the batch's prebuilt inverted index (word_index.h).
static int word_grad_segsum(sert_model* m, const DataSplit& ds, int64_t batch_index,
                            const float* src, float divisor) {
    const int d = m->cfg.word_dim;
    if ((size_t)batch_index >= ds.idx_batches.size()) SERT_FAIL("batch has no word index");
    const BatchIndex& bx = ds.idx_batches[(size_t)batch_index];
    unsigned char* touched = nullptr;   // (row flags are static per batch: DataSplit::idx_touched_bits)
    static const bool no_fused_upper = variant_knob("SERT_SEG_NO_FUSED_UPPER") != nullptr;   // cross-check knob
    const bool fused_upper = !no_fused_upper && bx.fused_upper_ok && ds.idx_heavy && d % 4 == 0 &&
                             bx.nlevels == 3 && bx.item_cnt[1] > 0;
    // the batch's heavy words: one streaming pass over src for all of them (kernels_seg.h: segsum_heavy)
    // (vectorspace only: a loglinear index marks dense words for the V_e-wide per-word sums of dzu_from_dj; its
    // word gradient -- when it comes through here at all: SERT_LL_NODEDUP / the row-wise loss path -- has one source
    // row per TOKEN, not per batch row, and hpart is sized for V_e columns, not d_w)
    // Round 5: INSIDE the tree's launches where those are the 32-lane forms (kernels_seg.h: segsum_rows_plus) -- the streaming
    // pass beside level 0, its combine beside level 1; SERT_HEAVY_NO_FUSE (variants build) keeps the two launches in front.
    bool heavy_fused = false, heavy_combined = false;
    PlusJob hjob = PlusJob();
    if (bx.dense_cnt > 0 && is_vs(m) && d % 4 == 0) {
        const int B = m->cfg.batch_size, d4 = d / 4;
        static const bool no_fuse = variant_knob("SERT_HEAVY_NO_FUSE") != nullptr;
        const bool lpi32 = d4 <= 32 || (d4 > 64 && 64 * cdiv(d4, 64) > 32 * cdiv(d4, 32));
        const bool bundled = bx.bundle_cnt > 0 && ds.idx_bundles;
        heavy_fused = !no_fuse && lpi32 && bx.nlevels >= 1 && bx.item_cnt[0] > 0 && bx.row_groups == 1 && !bundled;
        const uint4* cnt = ds.idx_dense_counts + (size_t)batch_index * B;
        if (heavy_fused) {
            hjob.kind = 1; hjob.rpb = heavy_rows_fused(B); hjob.extra = cdiv(B, hjob.rpb); hjob.src = src; hjob.cnt16 = cnt; hjob.part = m->hpart;
            hjob.words = (const int32_t*)ds.idx_dense_words + (size_t)batch_index * kHeavyMax;
            hjob.nheavy = bx.dense_cnt; hjob.nblocks = hjob.extra; hjob.B = B;
        } else {
        const int nblk = cdiv(B, kHeavyRowsPerBlock);
        const size_t lds = (size_t)4 * kHeavyMax * 32 * sizeof(float4);   // 32 KB
        hipLaunchKernelGGL(segsum_heavy, dim3(nblk * cdiv(d4, 32)), dim3(1024), lds, m->stream, src, cnt, B, d, m->hpart);
        hipLaunchKernelGGL(segsum_heavy_combine, dim3(bx.dense_cnt, cdiv(d4, 32)), dim3(256), 0, m->stream,
                           (const float*)m->hpart, nblk, d, (const int32_t*)ds.idx_dense_words + (size_t)batch_index * kHeavyMax,
                           bx.dense_cnt, m->g_rw, divisor);
        }
    }
    auto heavy_combine_job = [&]() { PlusJob j = hjob; j.kind = 2; j.extra = bx.dense_cnt; j.src = m->hpart; return j; };
    for (int l = 0; l < bx.nlevels; ++l) {
        const int nitems = bx.item_cnt[l];
        if (nitems == 0) continue;
        if (fused_upper && l == 1) {
            // levels 1 and 2 in one launch (kernels_seg.h: segsum_upper_fused)
            const int nb_normal = cdiv(nitems, 32);
            hipLaunchKernelGGL(segsum_upper_fused, dim3(nb_normal + bx.heavy_cnt, cdiv(d / 4, 32)), dim3(1024), 0, m->stream,
                               m->wpart + (size_t)bx.part_off[0] * d, ds.idx_items + bx.item_off[1], nitems, nb_normal,
                               ds.idx_heavy + bx.heavy_off, m->g_rw, d, divisor);
            break;
        }
        const float* in = (l == 0) ? src : m->wpart + (size_t)bx.part_off[l - 1] * d;
        const int32_t* rows = (l == 0) ? ds.idx_rows + bx.rows_off : nullptr;
        const int4* items = ds.idx_items + bx.item_off[l];
        float* pout = m->wpart + (size_t)bx.part_off[l] * d;
        if (heavy_fused && l <= 1) {
            // level 0 + the heavy words' partial sums / level 1 + their combine: one launch each
            PlusJob j = l == 0 ? hjob : heavy_combine_job();
            j.slot_is_row = (l == 0 && bx.slot_is_row) ? 1 : 0;
            int ni = nitems;
            // timing knock-outs (variants build, WRONG results; tools/experiments/r05_plus_ko.sh): 1 = the tree alone, 2 = the stream alone
            static const int ko = variant_knob("SERT_KO_PLUS") ? atoi(variant_knob("SERT_KO_PLUS")) : 0;
            if (ko == 1 && l == 0) j.extra = 0;
            if (ko == 2 && l == 0) ni = 0;
            hipLaunchKernelGGL(segsum_rows_plus, dim3(j.extra + cdiv(ni, 8), cdiv(d / 4, 32)), dim3(256), 0, m->stream, in, rows, items,
                               ni, m->g_rw, pout, d, divisor, j);
            if (l == 1) heavy_combined = true;
            continue;
        }
        // level 0 in bundles of short items (kernels_seg.h: segsum_rows_bundled; opt-in, SERT_SEG_BUNDLE=1 at upload -- the
        // same sums bit for bit as one item per lane group, tests/test_gpu_parity.py::test_word_gradient_bundled_level0)
#ifdef SERT_VARIANTS
        if (l == 0 && d % 4 == 0 && bx.bundle_cnt > 0 && ds.idx_bundles && bx.row_groups == 1) {
            hipLaunchKernelGGL(segsum_rows_bundled, dim3(cdiv(bx.bundle_cnt, 8), cdiv(d / 4, 32)), dim3(256), 0, m->stream, in, rows,
                               items, (const int32_t*)ds.idx_bundles + bx.bundle_off, (int)bx.bundle_cnt, m->g_rw, pout, d, divisor);
            continue;
        }
#endif
        if (d % 4 == 0) {
            // lane groups of 32 or 64 float4 columns, whichever wastes fewer lanes (d = 300: 75 chunks are
            // 3 x 32 at 78 % instead of 2 x 64 at 59 %: 147 -> 143 us at C4; the sums do not depend on it)
            const int ch = d / 4;
            const bool seg32y = ch > 64 && ch * (64 * cdiv(ch, 64)) > ch * (32 * cdiv(ch, 32));
            // row-grouped level 0: eight item lists, one per XCD (kernels_seg.h: XcdLists)
            XcdLists xl = XcdLists();
            xl.slot_is_row = (l == 0 && bx.slot_is_row) ? 1 : 0;
            int longest = 0;
            if (l == 0 && bx.row_groups > 1) {
                xl.on = 1;
                for (int x = 0; x < 8; ++x) { xl.off[x] = bx.xcd_off[x]; xl.cnt[x] = bx.xcd_cnt[x]; longest = std::max(longest, bx.xcd_cnt[x]); }
            }
            const int ipb = (d / 4 <= 32 || seg32y) ? 8 : 4;
            const int gx = xl.on ? 8 * cdiv(longest, ipb) : cdiv(nitems, ipb);
            if (d / 4 <= 32)
                hipLaunchKernelGGL((segsum_rows<32>), dim3(gx), dim3(256), 0, m->stream,
                                   in, rows, items, nitems, m->g_rw, pout, d, divisor, touched, 1, nullptr, nullptr, DenseSlots(), xl);
            else if (seg32y)
                hipLaunchKernelGGL((segsum_rows<32>), dim3(gx, cdiv(d / 4, 32)), dim3(256), 0, m->stream,
                                   in, rows, items, nitems, m->g_rw, pout, d, divisor, touched, 1, nullptr, nullptr, DenseSlots(), xl);
            else   // rows wider than 64 float4 chunks (d = 300: 75): the rest goes to further column groups
                hipLaunchKernelGGL((segsum_rows<64>), dim3(gx, cdiv(d / 4, 64)), dim3(256), 0, m->stream,
                                   in, rows, items, nitems, m->g_rw, pout, d, divisor, touched, 1, nullptr, nullptr, DenseSlots(), xl);
        } else {
            hipLaunchKernelGGL((segsum_rows_scalar<false>), dim3(cdiv(nitems, 4), cdiv(d, 64)), dim3(256), 0, m->stream, in,
                               rows, items, nitems, m->g_rw, pout, d, divisor, touched);
        }
    }
    if (heavy_fused && !heavy_combined) {   // (no level 1, or the fused upper levels took it: the combine alone)
        const PlusJob j = heavy_combine_job();
        hipLaunchKernelGGL(segsum_rows_plus, dim3(j.extra, cdiv(d / 4, 32)), dim3(256), 0, m->stream, (const float*)nullptr,
                           (const int32_t*)nullptr, (const int4*)nullptr, 0, m->g_rw, (float*)nullptr, d, divisor, j);
    }
    return 0;
}

// Loglinear on the distinct-word tables (kernels_ll.h): per word, the sum of the dJ rows of the
// batch rows it occurs in and the sum of its r_ik, then dZu = mask dJsum - P rsum.  Both sums
// ride the word's occurrence tree -- the same order-fixed tree as the word gradient, over
// V_e-wide rows (source row = batch row of the occurrence), destination = the word's rank
// among the batch's distinct words.
static int dzu_from_dj(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const int V = m->cfg.num_entities;
    const BatchIndex& bx = ds.idx_batches[(size_t)batch_index];
    // odd V_e: the scalar sums and the finishing expression ride the V_e-wide launches (kernels_seg.h: segsum_rows_scalar<true, true>)
    static const bool split_odd = variant_knob("SERT_LL_DZU_SPLIT") != nullptr;   // the three extra launches, for A/B
    const bool fused_odd = V % 4 != 0 && !split_odd;
    // the scalars first: the V_e-wide pass applies them when it stores a word's final row
    for (int l = 0; l < bx.nlevels && !fused_odd; ++l) {
        const int nitems = bx.item_cnt[l];
        if (nitems == 0) continue;
        const int32_t* rows = (l == 0) ? ds.idx_rows + bx.rows_off : nullptr;
        const int4* items = ds.idx_items + bx.item_off[l];
        const float* in = (l == 0) ? m->ll_r : m->ll_rpart + (size_t)bx.part_off[l - 1];
        float* pout = m->ll_rpart + (size_t)bx.part_off[l];
        hipLaunchKernelGGL(segsum_scalar_wave, dim3(cdiv(nitems, 4)), dim3(256), 0, m->stream, in, rows, items, nitems,
                           m->ll_rsum, pout);
    }
    // the dense heavy words inside the tree's launches (round 5; SERT_HEAVY_NO_FUSE in a variants build: the two launches behind the tree)
    static const bool ll_no_fuse = variant_knob("SERT_HEAVY_NO_FUSE") != nullptr;
    const bool ll_heavy_fused = !ll_no_fuse && V % 4 == 0 && bx.dense_cnt > 0 && bx.nlevels >= 1 && bx.item_cnt[0] > 0;
    bool ll_heavy_combined = false;
    PlusJobLL ll_job = PlusJobLL();
    if (ll_heavy_fused) {
        const int B = m->cfg.batch_size;
        ll_job.j.kind = 1; ll_job.j.rpb = heavy_rows_fused(B); ll_job.j.extra = cdiv(B, ll_job.j.rpb); ll_job.j.src = m->J;
        ll_job.j.cnt16 = ds.idx_dense_counts + (size_t)batch_index * B; ll_job.j.part = m->hpart;
        ll_job.j.nheavy = bx.dense_cnt; ll_job.j.nblocks = ll_job.j.extra; ll_job.j.B = B;
        ll_job.logp = m->Zu; ll_job.rsum = m->ll_rsum;
    }
    for (int l = 0; l < bx.nlevels; ++l) {
        const int nitems = bx.item_cnt[l];
        if (nitems == 0) continue;
        const int32_t* rows = (l == 0) ? ds.idx_rows_div + bx.rows_off : nullptr;   // batch row of the entry
        const int4* items = ds.idx_items + bx.item_off[l];
        const float* in = (l == 0) ? m->J : m->zpart + (size_t)bx.part_off[l - 1] * V;
        float* pout = m->zpart + (size_t)bx.part_off[l] * V;
        if (V % 4 == 0 && bx.dense_cnt > 0) {
            // (the batch's heavy words are summed densely: their items are skipped)
            DenseSlots dsl;
            dsl.n = bx.dense_cnt;
            for (int h = 0; h < kHeavyMax; ++h) dsl.slot[h] = h < bx.dense_cnt ? bx.dense_slot[h] : -2;
            if (ll_heavy_fused && l <= 1) {
                // ... by extra workgroups of this very launch (kernels_seg.h: segsum_rows_plus_ll): the stream over dJ beside
                // level 0, its combine and finishing expression beside level 1
                PlusJobLL j = ll_job;
                j.dense = dsl;
                if (l == 1) { j.j.kind = 2; j.j.extra = bx.dense_cnt; j.j.src = m->hpart; ll_heavy_combined = true; }
                hipLaunchKernelGGL(segsum_rows_plus_ll, dim3(j.j.extra + cdiv(nitems, 4), cdiv(V / 4, 64)), dim3(256), 0, m->stream, in,
                                   rows, items, nitems, m->dZu, pout, V, dsl, j);
                continue;
            }
            hipLaunchKernelGGL((segsum_rows<64, true, true, true>), dim3(cdiv(nitems, 4), cdiv(V / 4, 64)), dim3(256), 0,
                               m->stream, in, rows, items, nitems, m->dZu, pout, V, 1.0f,
                               (unsigned char*)nullptr, 1, (const float*)m->Zu,
                               (const float*)m->ll_rsum, dsl);
        } else if (V % 4 == 0) {
            hipLaunchKernelGGL((segsum_rows<64, true, true>), dim3(cdiv(nitems, 4), cdiv(V / 4, 64)), dim3(256), 0,
                               m->stream, in, rows, items, nitems, m->dZu, pout, V, 1.0f,
                               (unsigned char*)nullptr, 1, (const float*)m->Zu,
                               (const float*)m->ll_rsum);
        } else if (fused_odd) {
            hipLaunchKernelGGL((segsum_rows_scalar<true, true>), dim3(cdiv(nitems, 4), cdiv(V, 64)), dim3(256), 0, m->stream, in,
                               rows, items, nitems, m->dZu, pout, V, 1.0f, (unsigned char*)nullptr, 1, (const float*)m->Zu,
                               (const float*)((l == 0) ? m->ll_r : m->ll_rpart + (size_t)bx.part_off[l - 1]),
                               (l == 0) ? (const int32_t*)(ds.idx_rows + bx.rows_off) : (const int32_t*)nullptr,
                               m->ll_rsum, m->ll_rpart + (size_t)bx.part_off[l]);
        } else {
            hipLaunchKernelGGL((segsum_rows_scalar<true>), dim3(cdiv(nitems, 4), cdiv(V, 64)), dim3(256), 0, m->stream, in,
                               rows, items, nitems, m->dZu, pout, V, 1.0f, (unsigned char*)nullptr, 1);
        }
    }
    if (ll_heavy_fused && !ll_heavy_combined) {   // (no level 1: the combine alone)
        PlusJobLL j = ll_job;
        j.dense.n = bx.dense_cnt;
        for (int h = 0; h < kHeavyMax; ++h) j.dense.slot[h] = h < bx.dense_cnt ? bx.dense_slot[h] : -2;
        j.j.kind = 2; j.j.extra = bx.dense_cnt; j.j.src = m->hpart;
        hipLaunchKernelGGL(segsum_rows_plus_ll, dim3(j.j.extra, cdiv(V / 4, 64)), dim3(256), 0, m->stream, (const float*)nullptr,
                           (const int32_t*)nullptr, (const int4*)nullptr, 0, m->dZu, (float*)nullptr, V, j.dense, j);
    }
    if (V % 4 == 0 && bx.dense_cnt > 0 && !ll_heavy_fused) {
        // Heavy words: sum_i cnt[i][h] dJ[i, :] by ONE pass over dJ (262 MB at C2 dims) instead of one 4 kB
        // row fetch per occurrence -- a dozen words hold over half of a Zipfian batch's tokens
        const int B = m->cfg.batch_size, d4 = V / 4;
        const int nblk = cdiv(B, kHeavyRowsPerBlock);
        const uint4* cnt = ds.idx_dense_counts + (size_t)batch_index * B;
        static const bool attr_set = hipFuncSetAttribute((const void*)segsum_heavy_wide, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         2 * kHeavyMax * 128 * (int)sizeof(float4)) == hipSuccess;
        if (!attr_set) SERT_FAIL("cannot reserve the LDS of segsum_heavy_wide");
        hipLaunchKernelGGL(segsum_heavy_wide, dim3(nblk * cdiv(d4, 128)), dim3(512), (size_t)2 * kHeavyMax * 128 * sizeof(float4), m->stream,
                           (const float*)m->J, cnt, B, V, m->hpart);
        DenseSlots dsl;
        dsl.n = bx.dense_cnt;
        for (int h = 0; h < kHeavyMax; ++h) dsl.slot[h] = h < bx.dense_cnt ? bx.dense_slot[h] : -2;
        hipLaunchKernelGGL(segsum_heavy_combine_ll, dim3(bx.dense_cnt, cdiv(d4, 32)), dim3(256), 0, m->stream,
                           (const float*)m->hpart, nblk, V, dsl, m->dZu, (const float*)m->Zu, (const float*)m->ll_rsum);
    }
    if (V % 4 != 0 && !fused_odd)   // (odd V_e, split form: separate finishing pass)
        hipLaunchKernelGGL(ll_dzu_combine, dim3(grid_for((int64_t)m->ll_U * V)), dim3(256), 0, m->stream, m->dZu,
                           (const float*)m->Zu, (const float*)m->ll_rsum, (int64_t)m->ll_U, V);
    return 0;
}

// Stable sort of the (entity id, pair index) keys of this step: cand -> cand_sorted,
// iota -> pair_sorted (kernels_sort.h), LSD over ceil(bits/11) digits.
static int entity_key_sort(sert_model* m, int total, hipStream_t st, const int32_t* keys = nullptr) {
    const int tiles = cdiv(total, kSortTile);
    const int bits = m->sort_bits;
    const int passes = cdiv(bits, kSortMaxBits);
    const int width = cdiv(bits, passes);
    const int32_t* kin = keys ? keys : m->cand;
    const int32_t* vin = nullptr;  // value of element i is i
    for (int p = 0; p < passes; ++p) {
        const int shift = p * width;
        const int nb = std::min(width, bits - shift);
        const bool to_final = ((passes - 1 - p) % 2) == 0;
        int32_t* kout = to_final ? m->cand_sorted : m->sort_k_tmp;
        int32_t* vout = to_final ? m->pair_sorted : m->sort_v_tmp;
        // (first digit: also clears run_start / run_end, which the chunked reduce only writes for entities it meets --
        //  a step whose negatives were drawn ahead has no prologue launch to clear them)
        hipLaunchKernelGGL(csort_hist, dim3(tiles), dim3(256), 0, st, kin, total, shift, 1 << nb,
                           tiles, m->sort_hist, p == 0 ? m->run_start : (int32_t*)nullptr,
                           p == 0 ? (int)(2 * round_up(m->cfg.num_entities, 4)) : 0);
        hipLaunchKernelGGL(csort_scan_bins, dim3(cdiv(1 << nb, 4)), dim3(256), 0, st,
                           m->sort_hist, 1 << nb, tiles, m->sort_bin_total);
        hipLaunchKernelGGL(csort_scatter, dim3(tiles), dim3(256), 0, st, kin, vin, kout, vout,
                           total, shift, nb, tiles, m->sort_hist, m->sort_bin_total);
        kin = kout;
        vin = vout;
    }
    return 0;
}

// C (M,N, contiguous) = op(A).op(B) for a contraction over a LONG K (the entity
// vocabulary in the dX GEMMs of the full-softmax models) whose output has too few
// 128x128 tiles to fill 256 CUs: split K so that ~1024 workgroup items exist, then an
// order-fixed combine (deterministic).
template <bool TA, bool TB>
static int gemm_long_k(sert_model* m, hipStream_t s, const float* A, const float* Bm, float* C, int M,
                       int N, int K, int lda, int ldb, const int32_t* rowmap = nullptr, float* mapped_C = nullptr,
                       bool* mapped = nullptr) {
    if (mapped) *mapped = false;
    const int tiles = cdiv(M, GM) * cdiv(N, GN);
    int splits = 1;
    if (tiles < 512 && K >= 4096) splits = std::min(cdiv(1024, tiles), K / 1024);
    else if (tiles < 256 && K >= 512) splits = std::min(cdiv(1024, tiles), K / 128);   // few tiles, medium K
    if (!TA && K >= 4096 && gemm_x3_enabled()) {
        // the bf16-pipe kernel (gemm_x3.h) has larger tiles -- 256 rows, up to 320 columns --: enough k ranges for two
        // workgroups per CU (the loglinear dG over 100 000 entities: 9 tiles x 57 ranges; 1.83 -> 1.1 ms)
        const int t3 = cdiv(M, N <= 128 ? 128 : 256) * cdiv(N, N <= 128 ? 128 : 320);
        const int s3 = std::max(1, std::min(cdiv(512, t3), K / 1024));
        if (s3 > 1 && x3_shape_ok(false, TB, A, Bm, M, N, K, lda, ldb, s3)) splits = s3;
    }
    // (between one and two tiles per CU -- the loglinear dG at C2 dims, 347 tiles -- a three-way split was
    //  tried: 162 -> 175 us with its combine; co-resident workgroups share the matrix pipe without loss)
    if (splits <= 1) {
        launch_gemm<TA, TB, EPI_STORE>(s, A, Bm, C, nullptr, M, N, K, lda, ldb, N, 1, 0, 0, nullptr, nullptr, 0, rowmap, mapped_C,
                                       mapped);
        return 0;
    }
    const int kper = (int)round_up(cdiv(K, splits), GK);
    splits = cdiv(K, kper);
    const size_t mn = (size_t)M * N;
    if (mn * splits > m->skbuf_count) {
        SERT_HIP(hipStreamSynchronize(s));
        if (m->skbuf) (void)hipFree(m->skbuf);
        m->skbuf = nullptr;
        m->skbuf_count = mn * splits;
        SERT_TRY(dmalloc(&m->skbuf, m->skbuf_count));
    }
    launch_gemm<TA, TB, EPI_STORE>(s, A, Bm, m->skbuf, nullptr, M, N, K, lda, ldb, N, splits, kper, mn);
    if (rowmap && mapped_C) {
        // (the combine stores row r as row rowmap[r] of mapped_C: no copy kernel behind it)
        launch_reduce_partials(s, m->skbuf, splits, mn, mn, mapped_C, mn, mapped_C, rowmap, N);
        if (mapped) *mapped = true;
    } else launch_reduce_partials(s, m->skbuf, splits, mn, mn, C, mn, C);
    return 0;
}

// ---- data-parallel exchange (new: the reference is single-device) ---------
// ZeRO-1 over the big tensors (model.h): reduce-scatter of a gradient slab -> the optimiser on
// the owned piece (1/world of the launch, 1/world of the state) -> all-gather of the parameter
// slab; the small tensors' gradients, the loss sum and the owned pieces' sum of squares travel
// in ONE all-reduce and the small tensors are updated identically on every rank.  Every
// collective is issued on comm_stream, in the same order on every rank:
//   RS(R_w) as soon as the segmented reduction has produced dR_w (it overlaps the rest of the
//   backward), RS(other big tensors), AR(rest), then AG(R_w), AG(others) behind the optimiser.
static inline bool is_dp(const sert_model* m) { return m->comm != nullptr || m->host_ar != nullptr; }

struct ParamTensor {
    float *p, *g, *s0, *s1;
    size_t n;
    bool l2;
};
static ParamTensor param_tensor(sert_model* m, int i) {
    switch (i) {
        case 0: return {m->rw, m->g_rw, m->s0_rw, m->s1_rw, m->n_rw, true};
        case 1: return {m->re, m->g_re, m->s0_re, m->s1_re, m->n_re, true};
        case 2: return {m->W, m->g_w, m->s0_w, m->s1_w, m->n_w, true};
        default: return {m->b, m->g_b, m->s0_b, m->s1_b, m->n_b, false};   // bias: not regularised
    }
}
static inline size_t slab_elems(const sert_model* m, int i) { return m->pt_sc[i] * (size_t)m->world; }
static inline size_t piece_off(const sert_model* m, int i, int c) {
    return (size_t)c * slab_elems(m, i) + (size_t)m->rank * m->pt_sc[i];
}

// ---- host-mediated transport: every collective through ONE all-to-all callback ------------------
static int host_reserve(sert_model* m, size_t send_floats, size_t recv_floats) {
    if (m->host_send_cap < send_floats) {
        if (m->host_send) (void)hipHostFree(m->host_send);
        m->host_send = nullptr; m->host_send_cap = 0;
        SERT_HIP(hipHostMalloc((void**)&m->host_send, std::max<size_t>(send_floats, 64) * sizeof(float), hipHostMallocDefault));
        m->host_send_cap = std::max<size_t>(send_floats, 64);
    }
    if (m->host_recv_cap < recv_floats) {
        if (m->host_recv) (void)hipHostFree(m->host_recv);
        m->host_recv = nullptr; m->host_recv_cap = 0;
        SERT_HIP(hipHostMalloc((void**)&m->host_recv, std::max<size_t>(recv_floats, 64) * sizeof(float), hipHostMallocDefault));
        m->host_recv_cap = std::max<size_t>(recv_floats, 64);
    }
    return 0;
}
static int host_call(sert_model* m, const std::vector<int64_t>& soff, const std::vector<int64_t>& scnt,
                     const std::vector<int64_t>& roff, const std::vector<int64_t>& rcnt) {
    if (m->host_ar(m->host_ar_user, m->host_send, soff.data(), scnt.data(), m->host_recv, roff.data(), rcnt.data()) != 0)
        SERT_FAIL("host all-to-all callback failed");
    for (int q = 0; q < m->world; ++q)
        if (q != m->rank) m->comm_bytes_moved += 4.0 * (double)(scnt[(size_t)q] + rcnt[(size_t)q]);
    return 0;
}
// In-place SUM of dev[0, count) over the ranks (every rank sends the whole buffer to every rank and
// adds the copies in rank order: identical bits everywhere).  For the small replicated remainder.
static int host_allreduce(sert_model* m, float* dev, size_t count, hipStream_t st) {
    if (count == 0) return 0;
    const size_t W = (size_t)m->world;
    SERT_TRY(host_reserve(m, count, count * W));
    SERT_HIP(hipMemcpyAsync(m->host_send, dev, count * sizeof(float), hipMemcpyDeviceToHost, st));
    SERT_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> soff(W, 0), scnt(W, (int64_t)count), roff(W), rcnt(W, (int64_t)count);
    for (size_t q = 0; q < W; ++q) roff[q] = (int64_t)(q * count);
    SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
    float* acc = m->host_send;
    for (size_t k = 0; k < count; ++k) acc[k] = m->host_recv[k];
    for (size_t q = 1; q < W; ++q) {
        const float* src = m->host_recv + q * count;
        for (size_t k = 0; k < count; ++k) acc[k] += src[k];
    }
    SERT_HIP(hipMemcpyAsync(dev, acc, count * sizeof(float), hipMemcpyHostToDevice, st));
    SERT_HIP(hipStreamSynchronize(st));
    return 0;
}
// Reduce-scatter of one slab (world pieces of sc floats at `slab`): piece q goes to rank q, this rank
// adds the world copies of ITS piece in rank order and stores the sum over its piece -- only there.
static int host_reduce_scatter(sert_model* m, float* slab, size_t sc, hipStream_t st) {
    const size_t W = (size_t)m->world;
    SERT_TRY(host_reserve(m, sc * W, sc * W));
    SERT_HIP(hipMemcpyAsync(m->host_send, slab, sc * W * sizeof(float), hipMemcpyDeviceToHost, st));
    SERT_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> soff(W), scnt(W, (int64_t)sc), roff(W), rcnt(W, (int64_t)sc);
    for (size_t q = 0; q < W; ++q) { soff[q] = (int64_t)(q * sc); roff[q] = (int64_t)(q * sc); }
    SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
    float* acc = m->host_send;
    for (size_t k = 0; k < sc; ++k) acc[k] = m->host_recv[k];
    for (size_t q = 1; q < W; ++q) {
        const float* src = m->host_recv + q * sc;
        for (size_t k = 0; k < sc; ++k) acc[k] += src[k];
    }
    SERT_HIP(hipMemcpyAsync(slab + (size_t)m->rank * sc, acc, sc * sizeof(float), hipMemcpyHostToDevice, st));
    SERT_HIP(hipStreamSynchronize(st));
    return 0;
}
// All-gather of one slab: this rank's piece goes to everyone, piece q arrives from rank q.
static int host_allgather(sert_model* m, float* slab, size_t sc, hipStream_t st) {
    const size_t W = (size_t)m->world;
    SERT_TRY(host_reserve(m, sc, sc * W));
    SERT_HIP(hipMemcpyAsync(m->host_send, slab + (size_t)m->rank * sc, sc * sizeof(float), hipMemcpyDeviceToHost, st));
    SERT_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> soff(W, 0), scnt(W, (int64_t)sc), roff(W), rcnt(W, (int64_t)sc);
    for (size_t q = 0; q < W; ++q) roff[q] = (int64_t)(q * sc);
    SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
    SERT_HIP(hipMemcpyAsync(slab, m->host_recv, sc * W * sizeof(float), hipMemcpyHostToDevice, st));
    SERT_HIP(hipStreamSynchronize(st));
    return 0;
}
// All-to-all of float segments between device buffers (offsets / counts in floats, peers only).
static int host_alltoallv(sert_model* m, const float* dsend, size_t stotal, const std::vector<int64_t>& soff,
                          const std::vector<int64_t>& scnt, float* drecv, size_t rtotal, const std::vector<int64_t>& roff,
                          const std::vector<int64_t>& rcnt, hipStream_t st) {
    SERT_TRY(host_reserve(m, stotal, rtotal));
    if (stotal) SERT_HIP(hipMemcpyAsync(m->host_send, dsend, stotal * sizeof(float), hipMemcpyDeviceToHost, st));
    SERT_HIP(hipStreamSynchronize(st));
    SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
    if (rtotal) SERT_HIP(hipMemcpyAsync(drecv, m->host_recv, rtotal * sizeof(float), hipMemcpyHostToDevice, st));
    SERT_HIP(hipStreamSynchronize(st));
    return 0;
}
static int rccl_alltoallv(sert_model* m, const float* dsend, const std::vector<int64_t>& soff, const std::vector<int64_t>& scnt,
                          float* drecv, const std::vector<int64_t>& roff, const std::vector<int64_t>& rcnt, hipStream_t st) {
    SERT_NCCL(g_rccl.GroupStart());
    // a failing Send / Recv must not leave the group open (every later collective of this communicator would be
    // queued into the dangling group and hang): remember the first error, always close the group, then fail
    int first = 0;
    const char* what = "";
    for (int q = 0; q < m->world && first == 0; ++q) {
        if (q == m->rank) continue;
        if (scnt[(size_t)q]) {
            first = g_rccl.Send(dsend + soff[(size_t)q], (size_t)scnt[(size_t)q], /*ncclFloat32*/ 7, q, m->comm, st);
            if (first) { what = "ncclSend"; break; }
        }
        if (rcnt[(size_t)q]) {
            first = g_rccl.Recv(drecv + roff[(size_t)q], (size_t)rcnt[(size_t)q], 7, q, m->comm, st);
            if (first) { what = "ncclRecv"; break; }
        }
        m->comm_bytes_moved += 4.0 * (double)(scnt[(size_t)q] + rcnt[(size_t)q]);
    }
    const int end = g_rccl.GroupEnd();
    if (first != 0 || end != 0) {
        m->comm_dead = true;      // (peers may be blocked in a half-issued all-to-all: nothing sane can follow)
        SERT_FAIL(std::string(first ? what : "ncclGroupEnd") + " failed in the row all-to-all: " +
                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(first ? first : end) : "rccl error"));
    }
    return 0;
}

// ---- the word table exchanged by rows (kernels_xchg.h) -------------------------------------------
static inline bool xr_async(const sert_model* m) { return m->comm && !m->timing.enabled; }

// One all-to-all of rows: `send_cnt[q]` rows to rank q out of xr_send (peer-major), `recv_cnt[q]` rows
// from rank q into xr_recv (peer-major), d_w floats each.
static int xr_alltoall(sert_model* m, const std::vector<int32_t>& send_cnt, const std::vector<int32_t>& recv_cnt, hipStream_t st) {
    const size_t W = (size_t)m->world;
    const int64_t d = m->cfg.word_dim;
    std::vector<int64_t> soff(W), scnt(W), roff(W), rcnt(W);
    int64_t so = 0, ro = 0;
    for (size_t q = 0; q < W; ++q) {
        soff[q] = so; scnt[q] = (int64_t)send_cnt[q] * d; so += scnt[q];
        roff[q] = ro; rcnt[q] = (int64_t)recv_cnt[q] * d; ro += rcnt[q];
    }
    if (m->host_ar) return host_alltoallv(m, m->xr_send, (size_t)so, soff, scnt, m->xr_recv, (size_t)ro, roff, rcnt, st);
    return rccl_alltoallv(m, m->xr_send, soff, scnt, m->xr_recv, roff, rcnt, st);
}

// PARAMETERS of the rows batch `b` touches: the owners pack and send them, this rank scatters what
// it receives into its copy of R_w.  Runs on the communication stream behind the word table's update.
static int xr_fetch_params(sert_model* m, int64_t b) {
    if (!m->xr_on || m->rw_full || m->xr_fetched_batch == b) return 0;
    if ((size_t)b >= m->xr->batches.size()) SERT_FAIL("batch has no row-exchange lists");
    const RowExchangeBatch& xb = m->xr->batches[(size_t)b];
    const int d4 = m->cfg.word_dim / 4;
    const bool async = xr_async(m);
    hipStream_t st = async ? m->comm_stream : m->stream;
    ScopedTimer tm(m, TG_ALLGATHER);
    if (async) SERT_HIP(hipStreamWaitEvent(st, m->ev_word_updated, 0));   // (the owners' rows are final)
    if (xb.serve_total)
        hipLaunchKernelGGL(xchg_pack_rows, dim3(grid_for((int64_t)xb.serve_total * d4)), dim3(256), 0, st, (const float*)m->rw,
                           (const int32_t*)m->xr_serve + xb.serve_off, xb.serve_total, d4, reinterpret_cast<float4*>(m->xr_send));
    SERT_TRY(xr_alltoall(m, xb.serve_cnt, xb.fetch_cnt, st));
    if (xb.fetch_total)
        hipLaunchKernelGGL(xchg_unpack_rows, dim3(grid_for((int64_t)xb.fetch_total * d4)), dim3(256), 0, st, m->rw,
                           (const int32_t*)m->xr_fetch + xb.fetch_off, xb.fetch_total, d4, reinterpret_cast<const float4*>(m->xr_recv));
    if (async) {
        SERT_HIP(hipEventRecord(m->ev_params_ready, st));
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_params_ready, 0));
    }
    m->xr_fetched_batch = b;
    return 0;
}

// GRADIENT rows of batch `b`: packed from this rank's dR_w, sent to their owners; the rows received
// are added to this rank's own in rank order into dR_w[owned rows].
static int xr_return_grads(sert_model* m, int64_t b) {
    const RowExchangeBatch& xb = m->xr->batches[(size_t)b];
    const int d4 = m->cfg.word_dim / 4;
    const bool async = xr_async(m);
    hipStream_t st = async ? m->comm_stream : m->stream;
    ScopedTimer tm(m, TG_REDUCE_SCATTER);
    if (async) {
        SERT_HIP(hipEventRecord(m->ev_grad_ready[0], m->stream));
        SERT_HIP(hipStreamWaitEvent(st, m->ev_grad_ready[0], 0));
    }
    if (xb.fetch_total)
        hipLaunchKernelGGL(xchg_pack_rows, dim3(grid_for((int64_t)xb.fetch_total * d4)), dim3(256), 0, st, (const float*)m->g_rw,
                           (const int32_t*)m->xr_fetch + xb.fetch_off, xb.fetch_total, d4, reinterpret_cast<float4*>(m->xr_send));
    SERT_TRY(xr_alltoall(m, xb.fetch_cnt, xb.serve_cnt, st));
    if (xb.nunion)
        hipLaunchKernelGGL(xchg_reduce_rows, dim3(grid_for((int64_t)xb.nunion * d4)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(m->xr_recv), (const int32_t*)m->xr_ptr + xb.ptr_off,
                           (const int32_t*)m->xr_ent + xb.ent_off, (const int32_t*)m->xr_union + xb.union_off, xb.nunion, d4, m->g_rw);
    if (async) {
        SERT_HIP(hipEventRecord(m->ev_rs_done[0][0], st));
        m->rs_issued[0] = true;
    }
    return 0;
}

static void xr_free_lists(sert_model* m) {
    m->xr_on = false;
    (void)hipFree(m->xr_serve); (void)hipFree(m->xr_fetch); (void)hipFree(m->xr_union); (void)hipFree(m->xr_ent);
    (void)hipFree(m->xr_ptr); (void)hipFree(m->xr_ubits); (void)hipFree(m->xr_send); (void)hipFree(m->xr_recv);
    m->xr_serve = m->xr_fetch = m->xr_union = m->xr_ent = m->xr_ptr = nullptr;
    m->xr_ubits = nullptr; m->xr_send = m->xr_recv = nullptr;
    delete m->xr;
    m->xr = nullptr;
}

// The exchange lists of a freshly uploaded training split: every rank's per-batch touched bitmaps are
// all-gathered (in groups of batches: at most 64 MB of bitmaps at a time), the lists derived from
// them on the host (kernels_xchg.h) and uploaded.  COLLECTIVE: every rank uploads its split.
static int xr_build_lists(sert_model* m, const std::vector<uint32_t>& bits, int64_t nb, int64_t bit_words) {
    xr_free_lists(m);
    if (!m->xr_mode || !m->pt_sharded[0]) return 0;      // (rank-invariant: every rank leaves here or none does)
    const size_t W = (size_t)m->world;
    hipStream_t s = m->stream;
    {
        // This function is collective, `nb` is a per-rank quantity: agree on it BEFORE anything can return early
        // (a rank whose shard held no complete batch used to skip the bitmap all-gather its peers blocked in).
        // Three exact small integers per rank as floats: nb (< 2^48 split in two) and the bitmap width.
        std::vector<float> mine = {(float)(nb & 0xffffff), (float)(nb >> 24), (float)bit_words}, all3(3 * W, 0.f);
        if (m->host_ar) {
            SERT_TRY(host_reserve(m, 3, 3 * W));
            memcpy(m->host_send, mine.data(), 3 * sizeof(float));
            std::vector<int64_t> soff(W, 0), scnt(W, 3), roff(W), rcnt(W, 3);
            for (size_t q = 0; q < W; ++q) roff[q] = (int64_t)(3 * q);
            const double moved = m->comm_bytes_moved;
            SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
            m->comm_bytes_moved = moved;
            memcpy(all3.data(), m->host_recv, 3 * W * sizeof(float));
        } else {
            float *src = nullptr, *dst = nullptr;
            SERT_TRY(dmalloc(&src, 3));
            if (dmalloc(&dst, 3 * W) != 0) { (void)hipFree(src); return -1; }
            hipError_t he = hipMemcpyAsync(src, mine.data(), 3 * sizeof(float), hipMemcpyHostToDevice, s);
            const int rc = he == hipSuccess ? g_rccl.AllGather(src, dst, 3, /*ncclFloat32*/ 7, m->comm, s) : 0;
            if (he == hipSuccess) he = hipMemcpyAsync(all3.data(), dst, 3 * W * sizeof(float), hipMemcpyDeviceToHost, s);
            if (he == hipSuccess) he = hipStreamSynchronize(s);
            (void)hipFree(src); (void)hipFree(dst);
            if (rc != 0) SERT_FAIL("ncclAllGather of the batch counts failed");
            SERT_HIP(he);
        }
        for (size_t q = 0; q < W; ++q) {
            const int64_t nbq = (int64_t)all3[3 * q] + ((int64_t)all3[3 * q + 1] << 24);
            if (nbq != nb || (nb > 0 && (int64_t)all3[3 * q + 2] != bit_words))
                SERT_FAIL("data parallel: rank " + std::to_string(q) + " uploaded " + std::to_string(nbq) + " complete batches, rank " +
                          std::to_string(m->rank) + " " + std::to_string(nb) + " -- every rank must upload the same number of rows "
                          "(sert_amd.distributed.shard_rows)");      // (raised on EVERY rank: they all see the same table)
        }
    }
    if (nb == 0) return 0;
    m->xr = new RowExchangeLists();
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(nb, ((int64_t)64 << 20) / (bit_words * 4 * (int64_t)W)));
    std::vector<uint32_t> all;
    for (int64_t b0 = 0; b0 < nb; b0 += group) {
        const int64_t gb = std::min(group, nb - b0);
        const size_t cnt = (size_t)(gb * bit_words);
        all.resize(W * cnt);
        const uint32_t* mine = bits.data() + (size_t)(b0 * bit_words);
        if (m->host_ar) {
            SERT_TRY(host_reserve(m, cnt, cnt * W));
            memcpy(m->host_send, mine, cnt * 4);
            std::vector<int64_t> soff(W, 0), scnt(W, (int64_t)cnt), roff(W), rcnt(W, (int64_t)cnt);
            for (size_t q = 0; q < W; ++q) roff[q] = (int64_t)(q * cnt);
            const double moved = m->comm_bytes_moved;
            SERT_TRY(host_call(m, soff, scnt, roff, rcnt));
            m->comm_bytes_moved = moved;   // (set-up traffic is not part of a step)
            memcpy(all.data(), m->host_recv, W * cnt * 4);
        } else {
            float *src = nullptr, *dst = nullptr;
            SERT_TRY(dmalloc(&src, cnt));
            if (dmalloc(&dst, cnt * W) != 0) { (void)hipFree(src); return -1; }
            hipError_t he = hipMemcpyAsync(src, mine, cnt * 4, hipMemcpyHostToDevice, s);
            const int rc = he == hipSuccess ? g_rccl.AllGather(src, dst, cnt, /*ncclFloat32: bits only*/ 7, m->comm, s) : 0;
            if (he == hipSuccess) he = hipMemcpyAsync(all.data(), dst, W * cnt * 4, hipMemcpyDeviceToHost, s);
            if (he == hipSuccess) he = hipStreamSynchronize(s);
            (void)hipFree(src); (void)hipFree(dst);
            if (rc != 0) SERT_FAIL("ncclAllGather of the touched-row bitmaps failed");
            SERT_HIP(he);
        }
        build_row_exchange(all.data(), m->world, m->rank, gb, bit_words, m->xr_rows_per_rank, m->cfg.vocab_size, *m->xr);
    }
    const RowExchangeLists& L = *m->xr;
    auto up = [&](int32_t** d, const std::vector<int32_t>& h) -> int {
        SERT_TRY(dmalloc(d, std::max<size_t>(1, h.size())));
        if (!h.empty()) SERT_HIP(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        return 0;
    };
    SERT_TRY(up(&m->xr_serve, L.serve_rows)); SERT_TRY(up(&m->xr_fetch, L.fetch_rows));
    SERT_TRY(up(&m->xr_union, L.union_rows)); SERT_TRY(up(&m->xr_ent, L.ent)); SERT_TRY(up(&m->xr_ptr, L.ptr));
    SERT_TRY(dmalloc(&m->xr_ubits, std::max<size_t>(1, L.union_bits.size())));
    if (!L.union_bits.empty())
        SERT_HIP(hipMemcpyAsync(m->xr_ubits, L.union_bits.data(), L.union_bits.size() * 4, hipMemcpyHostToDevice, s));
    const size_t buf = (size_t)std::max<int32_t>(1, L.max_xfer_rows) * (size_t)m->cfg.word_dim;
    SERT_TRY(dmalloc(&m->xr_send, buf));
    SERT_TRY(dmalloc(&m->xr_recv, buf));
    SERT_HIP(hipStreamSynchronize(s));
    m->xr_on = true;
    return 0;
}

// Make every row of R_w on this rank current (all-gather of the owned slabs).  COLLECTIVE.
static int ensure_full_rw(sert_model* m) {
    if (m->rw_full || !m->pt_sharded[0] || !is_dp(m)) { m->rw_full = true; return 0; }
    if (m->comm_dead) SERT_FAIL("the communicator of this data-parallel model was destroyed");
    const size_t sc = m->pt_sc[0], slab = sc * (size_t)m->world;
    if (m->comm_stream) SERT_HIP(hipStreamSynchronize(m->comm_stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    for (int c = 0; c < m->ar_chunks; ++c) {
        if (m->host_ar) SERT_TRY(host_allgather(m, m->rw + (size_t)c * slab, sc, m->stream));
        else SERT_NCCL(g_rccl.AllGather(m->rw + (size_t)c * slab + (size_t)m->rank * sc, m->rw + (size_t)c * slab, sc, 7, m->comm, m->stream));
    }
    SERT_HIP(hipStreamSynchronize(m->stream));
    m->rw_full = true;
    m->xr_fetched_batch = -1;
    return 0;
}

// Reduce-scatter of big tensor i's gradient (complete on the main stream at this point).
static int exchange_grad(sert_model* m, int i) {
    if (!is_dp(m) || !m->pt_sharded[i]) return 0;
    if (m->comm_dead) SERT_FAIL("the communicator of this data-parallel model was destroyed");
    const ParamTensor t = param_tensor(m, i);
    const size_t sc = m->pt_sc[i];
    m->rs_issued[i] = false;
    if (i == 0 && m->xr_on) return xr_return_grads(m, m->xr_batch);
    // verification transport: slab by slab, this rank receives and sums ITS piece only
    if (m->host_ar) {
        ScopedTimer tm(m, TG_REDUCE_SCATTER);
        for (int c = 0; c < m->ar_chunks; ++c) SERT_TRY(host_reduce_scatter(m, t.g + (size_t)c * slab_elems(m, i), sc, m->stream));
        return 0;
    }
    if (m->timing.enabled) {   // timing mode: serial, on the main stream
        ScopedTimer tm(m, TG_REDUCE_SCATTER);
        for (int c = 0; c < m->ar_chunks; ++c)
            SERT_NCCL(g_rccl.ReduceScatter(t.g + (size_t)c * slab_elems(m, i), t.g + piece_off(m, i, c), sc,
                                           /*ncclFloat32*/ 7, /*ncclSum*/ 0, m->comm, m->stream));
        return 0;
    }
    SERT_HIP(hipEventRecord(m->ev_grad_ready[i], m->stream));
    SERT_HIP(hipStreamWaitEvent(m->comm_stream, m->ev_grad_ready[i], 0));
    m->comm_bytes_moved += 8.0 * (double)sc * (double)(m->world - 1) * m->ar_chunks;   // (N-1) pieces out, (N-1) in per slab
    for (int c = 0; c < m->ar_chunks; ++c) {
        SERT_NCCL(g_rccl.ReduceScatter(t.g + (size_t)c * slab_elems(m, i), t.g + piece_off(m, i, c), sc, 7, 0,
                                       m->comm, m->comm_stream));
        SERT_HIP(hipEventRecord(m->ev_rs_done[i][c], m->comm_stream));
    }
    m->rs_issued[i] = true;
    return 0;
}
static int allreduce_word_grad(sert_model* m) { return exchange_grad(m, 0); }

// The other big tensors' reduce-scatters, then the all-reduce of the replicated remainder
// [small tensors' gradients | loss sum | owned sum of squares].
static int allreduce_rest(sert_model* m) {
    if (!is_dp(m)) return 0;
    if (m->dp_late_join) {
        // (every gradient of the side stream -- a sharded entity table's as well as the replicated remainder -- is
        //  complete before the communication stream touches it; the main stream meets them again behind the collectives)
        SERT_HIP(hipEventRecord(m->ev_join, m->stream2));
        SERT_HIP(hipStreamWaitEvent(m->comm_stream, m->ev_join, 0));
    }
    for (int i = 1; i < 4; ++i) SERT_TRY(exchange_grad(m, i));
    float* rest = m->gflat + m->rest_off;
    const size_t count = m->gflat_count - m->rest_off;
    if (m->host_ar) { ScopedTimer t(m, TG_ALLREDUCE); return host_allreduce(m, rest, count, m->stream); }
    m->comm_bytes_moved += 16.0 * (double)count * (double)(m->world - 1) / (double)m->world;   // ring all-reduce: 2 (N-1)/N out + in
    if (m->timing.enabled) {
        ScopedTimer t(m, TG_ALLREDUCE);
        SERT_NCCL(g_rccl.AllReduce(rest, rest, count, 7, 0, m->comm, m->stream));
        return 0;
    }
    SERT_HIP(hipEventRecord(m->ev_rest_ready, m->stream));
    SERT_HIP(hipStreamWaitEvent(m->comm_stream, m->ev_rest_ready, 0));
    SERT_NCCL(g_rccl.AllReduce(rest, rest, count, 7, 0, m->comm, m->comm_stream));
    SERT_HIP(hipEventRecord(m->ev_ar_done, m->comm_stream));
    // the main stream waits for it in optimizer_and_loss, behind the big tensors' updates
    return 0;
}

// Sum of squares of the pieces this rank owns (pre-update values: the L2 term of the loss the
// step returns, sert/models.py:773-791) into the scalar slot that rides in the all-reduce.
// Launched behind the zeroing of the gradient buffer, on the stream that did it.
static int owned_sum_of_squares(sert_model* m, hipStream_t st) {
    if (!is_dp(m)) return 0;
    int nparts = 0;
    for (int i = 0; i < 4; ++i) {
        if (!m->pt_sharded[i]) continue;
        const ParamTensor t = param_tensor(m, i);
        const size_t owned = m->pt_sc[i] * (size_t)m->ar_chunks;
        const int nb = (int)std::min<size_t>(kOptBlocks / 2, std::max<size_t>(1, owned / 1024));
        hipLaunchKernelGGL(sumsq_pieces, dim3(nb), dim3(256), 0, st, (const float*)t.p + (size_t)m->rank * m->pt_sc[i],
                           m->pt_sc[i], slab_elems(m, i), m->ar_chunks, m->sq_scratch + 4 * kOptBlocks + nparts);
        nparts += nb;
    }
    if (nparts) hipLaunchKernelGGL(partials_to_scalar, dim3(1), dim3(256), 0, st, m->sq_scratch + 4 * kOptBlocks, nparts, m->g_sq);
    return 0;
}

// ---- the vectorspace step -----------------------------------------------------
static int reduce_rowloss(sert_model* m, hipStream_t st);

static int vs_negatives(sert_model* m, const int64_t* negatives, uint64_t stream_pos, hipStream_t st) {
    const auto& c = m->cfg;
    const int64_t count = (int64_t)c.batch_size * c.num_negatives;
    if (count == 0) return 0;
    if (negatives) {
        // (parity path: the device sampler cannot produce an id outside [0, V_e))
        const int64_t Ve = c.num_entities;
        for (int64_t i = 0; i < count; ++i)
            if (negatives[i] < 0 || negatives[i] >= Ve) SERT_FAIL("negative sample out of range [0, num_entities)");
        SERT_HIP(hipMemcpyAsync(m->neg_stage, negatives, count * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(convert_i64_to_i32, dim3(grid_for(count)), dim3(256), 0, st,
                           m->neg_stage, m->neg, count);
    } else {
        const int64_t global_offset = (int64_t)m->rank * count;
        // Philox stream position: even = training draws, odd = evaluation draws
        // (the reference keeps two independent RandomStreams, models.py:745-752).
        hipLaunchKernelGGL(vs_sample_negatives, dim3(grid_for((count + 3) / 4)), dim3(256), 0,
                           st, m->neg, count, global_offset, (uint32_t)c.num_entities,
                           c.seed, stream_pos);
    }
    return 0;
}

// gather + mean-pool + projection: needs neither the negatives nor the gradient buffers
static int vs_project(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, dw = c.word_dim, de = c.entity_dim;
    const size_t row0 = (size_t)batch_index * B;
    // gather + mean-pool + projection in ONE launch where the shape allows it (kernels_proj.h: d_w, d_e <= 128, window <= 10):
    // the h and the t of the two launches below, bit for bit where they run gemm_x3.  OPT-IN, SERT_PROJ_FUSED=1 (read at
    // sert_create, VARIANTS BUILD ONLY since round 6): round 5 measured it slower than the two launches at C2 and equal at 8192
    // rows.  SERT_GEMM_FP32=1 (the fused kernel multiplies on the bf16 pipe) keeps the two launches too.
#ifdef SERT_VARIANTS
    if (m->proj_fused && gemm_x3_enabled() && vs_project_fused_ok(B, n, dw, de, m->n_rw)) {
        if (m->T_alt) std::swap(m->T, m->T_alt);     // (see below: this projection goes to the other buffer)
        ScopedTimer t(m, TG_GATHER);
        SERT_ID_DISPATCH(c.id_bytes, {
            const IdT* X = (const IdT*)ds.x + row0 * n;
            hipLaunchKernelGGL((vs_project_x3<IdT>), dim3(vs_project_grid(B, m->num_cus)), dim3(PJ_THREADS), 0, m->stream, X, (const float*)m->rw,
                               (const float*)m->W, (const float*)m->b, m->H, m->T, B, n, dw, de);
        });
        return 0;
    }
#endif
    {
        ScopedTimer t(m, TG_GATHER);
        SERT_ID_DISPATCH(c.id_bytes, {
            const IdT* X = (const IdT*)ds.x + row0 * n;
            // the batch's hot rows from LDS (kernels_vs.h: vs_gather_mean_hot; training batches with an index and dense words).
            // OPT-IN in a VARIANTS BUILD, SERT_GATHER_HOT=1 (read at upload: the slot bytes exist only then): same h bit for bit, measured SLOWER --
            // profiles/r05_experiments.txt, item 32.
#ifdef SERT_VARIANTS
            const int nhot = (ds.idx_tok_slot && (size_t)batch_index < ds.dense_cnt_of.size()) ? ds.dense_cnt_of[(size_t)batch_index] : 0;
            if (dw % 4 == 0 && nhot > 0 && (size_t)nhot * dw * sizeof(float) <= 48 * 1024)
                hipLaunchKernelGGL((vs_gather_mean_hot<IdT>), dim3(std::min<int64_t>(grid_for((int64_t)B * dw / 4, 256, 1 << 20), 8 * m->num_cus)),
                                   dim3(256), (size_t)nhot * dw * sizeof(float), m->stream, X, (const uint8_t*)ds.idx_tok_slot + row0 * n,
                                   (const int32_t*)ds.idx_dense_words + (size_t)batch_index * kHeavyMax, nhot, (const float*)m->rw, m->H, B, n, dw);
            else
#endif
            if (dw % 4 == 0)
                hipLaunchKernelGGL((vs_gather_mean<IdT, 4>), dim3(grid_for((int64_t)B * dw / 4, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, X, m->rw, m->H, B, n, dw);
            else
                hipLaunchKernelGGL((vs_gather_mean<IdT, 1>), dim3(grid_for((int64_t)B * dw, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, X, m->rw, m->H, B, n, dw);
        });
    }
    // The entity-gradient chain of the PREVIOUS step reads that step's projection rows on the side stream, and the main stream
    // no longer joins that stream at the end of a step (only the next LOSS kernel waits for it, settle_entity_update): this
    // projection goes to the other buffer.  (One buffer was a race the schedule merely kept from happening -- at C2 the
    // chain ends 15 us before the step does; small batches with the chain longer than the rest of the step lost it.)
    if (m->T_alt) std::swap(m->T, m->T_alt);
    {
        ScopedTimer t(m, TG_GEMM_FWD);
        // t = tanh(h.W + b)   (models.py:1057-1061)
#ifdef SERT_VARIANTS
        if (gemm_strip_ok(B, de, dw, dw, de, false, m->H, m->W))
            launch_gemm_strip<false, EPI_BIAS_TANH>(m->stream, m->H, m->W, m->T, m->b, B, de, dw, dw, de, de);
        else
#endif
            launch_gemm<false, false, EPI_BIAS_TANH>(m->stream, m->H, m->W, m->T, m->b, B, de, dw, dw,
                                                     de, de);
    }
    return 0;
}

// Events that mark the end of ONE kernel ride on that kernel's completion signal
// (SERT_EXT_EVENTS=0: plain hipEventRecord behind it, ~7 us of queue stall each).
static bool ext_events() {
    static const bool on = !(variant_knob("SERT_EXT_EVENTS") && atoi(variant_knob("SERT_EXT_EVENTS")) == 0);
    return on;
}

// Single GPU, two streams: ONE fork per step, behind the dh GEMM (the last reader of W): the side
// stream then takes the entity chain, dW / db and the small-tensor optimiser in a row with no
// further event, the main stream keeps loss -> dh -> segmented sum -> word-table optimiser.
// Every cross-queue event costs its queue ~5-7 us (the kernel that carries a completion signal
// ends with a cache write-back): two per step instead of three.  SERT_FORK_LATE=0 restores the
// fork right behind the NCE kernel with dW on the main stream.
static bool side_heavy_mode(const sert_model* m);
// Grid of dense_update_skip given the dense launches' grid for the same table (kOptBlocks, twice that from 2^24 elements).
static int skip_grid(int nb_dense, size_t /*elements*/) { return nb_dense; }

static bool fork_late_mode(const sert_model* m) {
    static const bool on = !(variant_knob("SERT_FORK_LATE") && atoi(variant_knob("SERT_FORK_LATE")) == 0);
    return on && !side_heavy_mode(m) && ext_events() && !is_dp(m) && !m->timing.enabled && m->nstreams == 2 &&
           m->n_re <= ((size_t)1 << 22) && m->cfg.kind == SERT_KIND_VECTORSPACE;
}

// Where the one fork of fork_late_mode sits: behind the dh GEMM (default) or, SERT_FORK_AT=nce,
// behind the NCE kernel -- W and b are updated on the main stream, so nothing on the side stream
// has to wait for the last reader of W any more, and the entity chain then runs beside the
// MFMA-bound dh / dW GEMMs instead of beside the cache-bound segmented sum.
static bool fork_at_nce(const sert_model* m) {
    static const bool on = variant_knob("SERT_FORK_AT") && !strncmp(variant_knob("SERT_FORK_AT"), "nce", 3);
    return on && fork_late_mode(m);
}
// SERT_FORK_AT=nce_dw: ... and the side stream starts with dW, db and the loss partials -- beside the dh GEMM (both
// 512-workgroup MFMA launches that leave half the matrix pipe idle on their own) -- and only then takes the entity chain,
// which then runs beside the segmented sum as in the default schedule.
static bool fork_at_nce_dw(const sert_model* m) {
    static const bool on = variant_knob("SERT_FORK_AT") && !strcmp(variant_knob("SERT_FORK_AT"), "nce_dw");
    return on && fork_at_nce(m);
}

// Late fork + a THIRD queue for the MFMA-bound dW GEMM, its combine and the W, b update: they only
// need da and h, so they can run beside the cache-bound segmented sum instead of in front of it.
// The queue waits on the same completion signal as the side stream (free for the main stream) and
// is joined in front of the loss finalisation.  SERT_DW_THIRD=0 keeps them on the main stream.
static bool dw_third_queue(const sert_model* m) {
    static const bool on = variant_knob("SERT_DW_THIRD") && atoi(variant_knob("SERT_DW_THIRD")) != 0;
    return on && fork_late_mode(m) && !fork_at_nce(m);   // (the W update must stay behind the dh GEMM)
}

// Single GPU, BIG entity table (more than 2^22 elements: the sorted entity-gradient chain and a streaming
// optimiser launch of its own -- C4): the main stream keeps nothing but the critical chain
//   loss -> dh GEMM -> segmented sum -> word-table optimiser -> tail,
// the side stream takes, forked on the loss kernel,
//   entity chain -> entity-table optimiser -> dW GEMM,
// and is joined in front of the tail.  The MFMA-bound dW (off the critical path: it only feeds the tail)
// and the 0.96 GB of the entity table's optimiser then run BESIDE the 750 us the word table streams,
// instead of in front of and behind it.  SERT_SIDE_HEAVY=0 restores dW in front of dh on the main stream
// and both optimiser launches behind the join.
static bool side_heavy_mode(const sert_model* m) {
    static const int level = knob("SERT_SIDE_HEAVY") ? atoi(knob("SERT_SIDE_HEAVY")) : 1;   // 2: small entity tables too
    return level > 0 && ext_events() && !is_dp(m) && !m->timing.enabled && m->nstreams == 2 && (m->pt_big[1] || level > 1) &&
           !m->pt_big[2] && m->cfg.kind == SERT_KIND_VECTORSPACE && !m->cfg.keep_grads;
}

// NCE score / loss / gradient coefficients
template <bool TRAIN>
static int vs_loss(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, de = c.entity_dim;
    const size_t row0 = (size_t)batch_index * B;
    {
        ScopedTimer t(m, TG_LOSS);
        const int32_t* y = ds.y + row0;
        const float* w = TRAIN ? ds.w + row0 : nullptr;
        const float inv_batch = 1.0f / (float)c.global_batch_size;
        dim3 block(256);
        // training with a side stream: the fork event of the backward pass is this kernel's own
        // completion signal (common.h: SERT_LAUNCH)
        m->fork_bound = false;
        if (de > 512) SERT_FAIL("entity_dim > 512 is not supported");   // (before an event is armed)
        if (TRAIN && ext_events() && (!fork_late_mode(m) || fork_at_nce(m)) && !m->timing.enabled && m->nstreams >= 2 && de % 4 == 0) {
            set_stop_event(m->ev_fork);
            m->fork_bound = true;
        }
        if (de % 4 == 0) {
            const int nch = cdiv(de / 4, 16);
            dim3 grid(cdiv(B, 16));
#define SERT_NCE_CASE(N)                                                                     \
    case N:                                                                                  \
        SERT_LAUNCH((vs_nce<N, TRAIN>), grid, block, 0, m->stream, m->T, m->re, y,          \
                           m->neg, w, m->DA, m->coef, m->cand, m->rowloss, B,               \
                           c.num_negatives, de, inv_batch, TRAIN ? m->red_loss : (float*)nullptr); \
        break;
#define SERT_NCE_REGS(N, C)                                                                  \
    SERT_LAUNCH((vs_nce_regs<N, TRAIN, C>), grid, block, 0, m->stream, m->T, m->re, y,       \
                       m->neg, w, m->DA, m->coef, m->cand, m->rowloss, B, c.num_negatives,   \
                       de, inv_batch, TRAIN ? m->red_loss : (float*)nullptr)
            static const bool no_regs = variant_knob("SERT_NCE_PER_CANDIDATE") != nullptr;
            const int nc = c.num_negatives + 1;
            // (d_e = 300, five float4 per lane and candidate: 256 VGPRs + AGPR spills, one wave per SIMD --
            //  191 us against 169 us for the per-candidate kernel at C4: the limit stays at four)
            if (!no_regs && nch <= 4 && nc <= 12) {
                // every candidate row of a row in registers (kernels_vs.h: vs_nce_regs)
                const int key = nch * 2 + (nc > 6 ? 1 : 0);
                switch (key) {
                    case 2: SERT_NCE_REGS(1, 6); break;
                    case 3: SERT_NCE_REGS(1, 12); break;
                    case 4: SERT_NCE_REGS(2, 6); break;
                    case 5: SERT_NCE_REGS(2, 12); break;
                    case 6: SERT_NCE_REGS(3, 6); break;
                    case 7: SERT_NCE_REGS(3, 12); break;
                    case 8: SERT_NCE_REGS(4, 6); break;
                    default: SERT_NCE_REGS(4, 12); break;
                }
            } else
            switch (nch) {
                SERT_NCE_CASE(1) SERT_NCE_CASE(2) SERT_NCE_CASE(3) SERT_NCE_CASE(4)
                SERT_NCE_CASE(5) SERT_NCE_CASE(6) SERT_NCE_CASE(7) SERT_NCE_CASE(8)
                default: SERT_FAIL("entity_dim > 512 is not supported");
            }
#undef SERT_NCE_REGS
#undef SERT_NCE_CASE
            set_stop_event(nullptr);   // (never leave an armed event behind a launch that did not happen)
            // (training: the kernel left one loss partial per workgroup in red_loss)
            m->nce_loss_partials = TRAIN ? cdiv(B, 16) : 0;
        } else {
            m->nce_loss_partials = 0;
            const int npl = cdiv(de, 64);
            dim3 grid(cdiv(B, 4));
#define SERT_NCE_CASE(N)                                                                     \
    case N:                                                                                  \
        hipLaunchKernelGGL((vs_nce_scalar<N, TRAIN>), grid, block, 0, m->stream, m->T,      \
                           m->re, y, m->neg, w, m->DA, m->coef, m->cand, m->rowloss, B,     \
                           c.num_negatives, de, inv_batch);                                  \
        break;
            switch (npl) {
                SERT_NCE_CASE(1) SERT_NCE_CASE(2) SERT_NCE_CASE(3) SERT_NCE_CASE(4)
                SERT_NCE_CASE(5) SERT_NCE_CASE(6) SERT_NCE_CASE(7) SERT_NCE_CASE(8)
                default: SERT_FAIL("entity_dim > 512 is not supported");
            }
#undef SERT_NCE_CASE
        }
    }
    return 0;
}

// dh and dW (+ db) of the projection in one launch (gemm_bwd_fused.h) where the shape allows it
static bool bwd_fused_applies(const sert_model* m) {
    // opt-in (SERT_BWD_FUSED=1): measured EQUAL to the two gemm.h launches at C2 (57.5 us against 29.3 + 29.2;
    // step 0.3030 against 0.3046 ms, inside the run-to-run spread) -- the fused kernel keeps the matrix pipe as
    // busy as they do (48 %), it only saves a launch and half of the partial slabs
#ifdef SERT_VARIANTS
    static const bool on = variant_knob("SERT_BWD_FUSED") && atoi(variant_knob("SERT_BWD_FUSED")) != 0;
    return on && m->cfg.kind == SERT_KIND_VECTORSPACE && m->cfg.word_dim == FB_D && m->cfg.entity_dim == FB_D &&
           m->cfg.batch_size >= 1024 && m->nstreams < 3 && (size_t)256 * (FB_D * FB_D + FB_D) <= m->part_count;
#else
    (void)m;
    return false;      // (the kernel lives in csrc/variants/gemm_bwd_fused.h: not in the product library)
#endif
}

// Few (pair, entity) keys over a table too large for the sort-free LDS path: the one-launch range kernel instead of
// sort + chunked reduce + fix-up (eight launches).  The scan costs ranges x pairs id reads: capped at 64 M (~256 MB out of L2).
// OPT-IN in a VARIANTS BUILD (SERT_EGRAD_RANGES=1 at sert_create): 32 us alone against 67 for the eight launches at the product-search settings,
// but the STEP does not move (0.202-0.207 against 0.199-0.203 ms: that chain is not what the step waits for; round 5).
#ifdef SERT_VARIANTS
static bool egrad_ranges_ok(const sert_model* m, int total) {
    const long long ranges = cdiv(m->cfg.num_entities, kERange);
    return m->egrad_ranges && !m->egrad_force_sort && !m->epart && m->cfg.entity_dim % 4 == 0 && total <= (1 << 20) && ranges * (long long)total <= (64ll << 20);
}
#endif

static int vs_backward(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, dw = c.word_dim, de = c.entity_dim;
    const size_t row0 = (size_t)batch_index * B;
    const bool fork_late = fork_late_mode(m);
    const bool fork_nce = fork_late && fork_at_nce(m);
    const bool side_heavy = side_heavy_mode(m);
    const bool fused_bwd = bwd_fused_applies(m) && !side_heavy;
#ifdef SERT_VARIANTS
    const int fused_grid = std::min(256, cdiv(B, FB_ROWS));   // one workgroup per CU, or per strip if there are fewer
#else
    const int fused_grid = 0;
#endif
    m->bucket_early = false;
    auto early_bucket = [&]() -> int {
        // Round 6: the PARTITION of this step's (pair, entity) keys by entity range (egrad_bucket, 19 us at C2) needs the
        // labels and the negatives only -- not the loss kernel's coefficients -- and this step's negatives were drawn on this
        // very stream during the previous step (neg_side_ready): it goes out in front of the fork as well and runs beside
        // gather / projection / loss.  The chain behind the fork is then egrad_acc alone: it starts 19 us earlier and ends
        // that much earlier beside the word table's update (profiles/r06_experiments.txt, item 1).
        static const bool no_early_bucket = variant_knob("SERT_NO_EARLY_BUCKET") != nullptr;
        m->bucket_early = false;
        // Measured (tools/experiments/r06_early_bucket.sh, r06_fork_nce_early.sh; three rounds each on one box, ms/step early / behind
        // the fork): batch 32768 0.1426-0.1466 / 0.1515-0.1552 (-5.5 %), 65536 0.2404-0.2427 / 0.2404-0.2426 (equal: egrad_acc ends
        // 29 us earlier, the tree beside it stretches by 5), 16384 0.1182-0.1213 / 0.1170-0.1184 (+1.5 %), 8192 0.0977-0.1000 /
        // 0.0938-0.0966 (+3-5 %: there the partition beside the forward delays the loss kernel and the update): from batch 32768.
        // SERT_EARLY_BUCKET=0 / 1 (variants build) forces it off / on.
        static const int early_knob = variant_knob("SERT_EARLY_BUCKET") ? atoi(variant_knob("SERT_EARLY_BUCKET")) : -1;
        const bool want_early = early_knob >= 0 ? early_knob != 0 : B >= 32768;
        if (!no_early_bucket && want_early && fork_late && !is_dp(m) && !m->timing.enabled && m->epart && c.kind == SERT_KIND_VECTORSPACE &&
            c.num_negatives > 0 && m->neg_side_ready && ds.y) {
            ScopedTimer t(m, TG_SORT, m->stream2);
            hipLaunchKernelGGL(egrad_bucket, dim3(m->eg_num_sub), dim3(512), 0, m->stream2, (const int32_t*)nullptr, B, c.num_negatives + 1,
                               m->eg_sub_rows, m->eg_er_shift, m->eg_ranges, m->eg_entries, m->eg_offs,
                               (const int32_t*)ds.y + row0, (const int32_t*)m->neg);
            m->bucket_early = true;
        }
        return 0;
    };
    // ... and the same for the SORTED entity chain of a larger entity table (V_e > 2048: the reference's product-search settings, C4):
    // the stable counting sort of the (entity, pair) keys -- six of the chain's eight launches -- needs the labels and the negatives
    // only.  With the negatives drawn ahead on this stream it goes out in front of the fork wait and runs beside gather / projection /
    // loss; behind the fork the chain is the chunked reduce + the fix-up.  (The first histogram pass also clears the per-entity run
    // bounds: nothing of the previous step reads them any more -- its fix-up precedes this in the stream.)
    m->sort_early = false;
    auto early_sort = [&]() -> int {
        // Measured (tools/experiments/r06_early_sort.sh, r06_early_sort_sizes.sh; two to three rounds each on one box; ms/step beside the
        // forward / inside the chain): the reference's product-search settings (batch 4096, V_e 32768, d_w 300) 0.1669-0.1689 / 0.1696-0.1727,
        // the same at batch 1024 0.1434-0.1443 / 0.1513-0.1527; d = 128, V_e 32768: batch 16384 0.1514-0.1529 / 0.1717-0.1727 (-12 %), 32768
        // 0.2034-0.2049 / 0.2255-0.2279, 65536 0.3227-0.3253 / 0.3434-0.3457; V_e 100000: batch 65536 at d = 128 0.4099-0.4140 / 0.4319-0.4355,
        // d = 300: batch 16384 0.705-0.714 / 0.709-0.717, 32768 0.865-0.920 / 0.906-0.954 -- but C4 itself (batch 65536, d = 300) 1.404-1.411 /
        // 1.360-1.364: there the chunked reduce (865 MB of row fetches) then starts beside the word gradient's tree (680 MB of them) instead of
        // beside the update, and the tree takes 389 us instead of 125.  Taken while dh, the tree's source, is below 64 MB.
        // SERT_EARLY_SORT=1 (variants build) forces it, SERT_NO_EARLY_SORT=1 switches it off.
        static const bool off = variant_knob("SERT_NO_EARLY_SORT") != nullptr;
        static const bool force = variant_knob("SERT_EARLY_SORT") && atoi(variant_knob("SERT_EARLY_SORT")) != 0;
        if (!force && (size_t)B * dw * sizeof(float) >= ((size_t)64 << 20)) return 0;
        if (off || m->epart || !m->cand_early || is_dp(m) || m->timing.enabled || m->nstreams < 2 || c.kind != SERT_KIND_VECTORSPACE ||
            c.num_negatives <= 0 || !m->neg_side_ready || !ds.y || c.num_entities <= 0)
            return 0;
        const int total = B * (c.num_negatives + 1);
        ScopedTimer t(m, TG_SORT, m->stream2);
        hipLaunchKernelGGL(vs_build_cand, dim3(grid_for(total)), dim3(256), 0, m->stream2, (const int32_t*)ds.y + row0, (const int32_t*)m->neg, B,
                           c.num_negatives, m->cand_early);
        SERT_TRY(entity_key_sort(m, total, m->stream2, m->cand_early));
        m->sort_early = true;
        return 0;
    };
    auto entity_grad = [&]() -> int {
        // fork: this chain only depends on the NCE kernel and is independent of the
        // GEMMs / word-table reduction below, so it runs on the side stream
        // (timing mode measures every kernel alone: everything stays on the main stream)
        hipStream_t st = (m->timing.enabled || m->nstreams < 2) ? m->stream : m->stream2;
        if ((!fork_late || fork_nce) && !m->dw_side_first) {   // (nce_dw: the side stream has met the fork already)
            if (!m->fork_bound) SERT_HIP(hipEventRecord(m->ev_fork, m->stream));
            m->fork_bound = false;
            if (st != m->stream) SERT_HIP(hipStreamWaitEvent(st, m->ev_fork, 0));
        }
        const int total = B * (c.num_negatives + 1);
        const int V = c.num_entities;
        m->re_in_parts = false;
        static const bool ko_egrad = variant_knob("SERT_KO_EGRAD") != nullptr;   // timing knock-out (wrong results)
        if (ko_egrad) {
        } else if (m->epart) {
            // small entity vocabulary: no global sort -- pairs bucketed by entity range per sub-group,
            // then row groups x entity ranges with the accumulators in LDS (kernels_egrad.h)
            const int de4 = de / 4, c1 = c.num_negatives + 1;
            const size_t lds = (size_t)4 * 16 * de * sizeof(float);
            const int grid = 8 * cdiv(m->eg_groups, 8) * m->eg_ranges;
            if (!m->bucket_early) {     // (else: the partition ran beside the forward, see dh_gemm below)
                ScopedTimer t(m, TG_SORT, st);
                hipLaunchKernelGGL(egrad_bucket, dim3(m->eg_num_sub), dim3(512), 0, st, m->cand, B, c1, m->eg_sub_rows,
                                   m->eg_er_shift, m->eg_ranges, m->eg_entries, m->eg_offs, (const int32_t*)nullptr,
                                   (const int32_t*)nullptr);
            }
            {
                ScopedTimer t(m, TG_EGRAD, st);
#define SERT_EL_ARGS m->eg_entries, m->eg_offs, m->coef, m->T, c1, de, V, m->eg_sub_rows, m->eg_num_sub, \
                     m->eg_subs_per_group, m->eg_groups, m->eg_ranges, m->epart
                hipLaunchKernelGGL((egrad_acc<2>), dim3(grid), dim3(256), lds, st, SERT_EL_ARGS);
#undef SERT_EL_ARGS
            }
            // Single GPU: the only reader of dR_e is the small-tensor optimiser, which adds the row
            // groups' tables itself (same order) -- no launch for the sum.  Data parallel: the
            // all-reduce needs the summed table.
            static const bool no_fold = variant_knob("SERT_EGRAD_GROUP_SUM") != nullptr;
            m->re_in_parts = !is_dp(m) && !m->pt_big[1] && !no_fold;
            if (!m->re_in_parts) {
                ScopedTimer t(m, TG_EFIX, st);
                const size_t table4 = (size_t)V * de4;
                hipLaunchKernelGGL(egrad_group_sum, dim3(grid_for((int64_t)table4)), dim3(256), 0, st, m->epart, m->eg_groups,
                                   table4, m->g_re);
            }
#ifdef SERT_VARIANTS
        } else if (egrad_ranges_ok(m, total)) {
            // few pairs over a mid-size table: one workgroup per range of 32 entities, no sort (variants/kernels_egrad_ranges.h)
            ScopedTimer t(m, TG_EGRAD, st);
            hipLaunchKernelGGL(egrad_ranges, dim3(cdiv(V, kERange)), dim3(256), 0, st, (const int32_t*)m->cand, (const float*)m->coef,
                               (const float*)m->T, total, c.num_negatives + 1, de, V, m->g_re);
#endif
        } else {
        // dR_e: stable sort of the (entity, pair) keys, chunked reduce, carry fix-up
        if (!m->sort_early) {       // (else: the keys were sorted beside the forward, see early_sort below)
            ScopedTimer t(m, TG_SORT, st);
            SERT_TRY(entity_key_sort(m, total, st));
        }
        const int chunks = cdiv(total, kEChunk);
        dim3 cgrid(cdiv(chunks, 16)), fgrid(cdiv(V, 4)), blk(256);
#define SERT_EG_ARGS m->cand_sorted, m->pair_sorted, m->coef, m->T, total, c.num_negatives + 1, de, \
                     m->g_re, m->ehead, m->etail, m->run_start, m->run_end
        {
            ScopedTimer t(m, TG_EGRAD, st);
            if (de % 4 == 0) {
                const int nch = cdiv(de / 4, 16);
                if (nch <= 1)      hipLaunchKernelGGL((egrad_chunk_reduce<4, 1>), cgrid, blk, 0, st, SERT_EG_ARGS);
                else if (nch <= 2) hipLaunchKernelGGL((egrad_chunk_reduce<4, 2>), cgrid, blk, 0, st, SERT_EG_ARGS);
                else if (nch <= 5) hipLaunchKernelGGL((egrad_chunk_reduce<4, 5>), cgrid, blk, 0, st, SERT_EG_ARGS);
                else               hipLaunchKernelGGL((egrad_chunk_reduce<4, 8>), cgrid, blk, 0, st, SERT_EG_ARGS);
            } else {
                hipLaunchKernelGGL((egrad_chunk_reduce<1, 4>), cgrid, blk, 0, st, SERT_EG_ARGS);
            }
        }
        {
            ScopedTimer t(m, TG_EFIX, st);
            const bool few = V < 256 && de <= 512;   // few entities: one workgroup per entity (long runs)
            if (de % 4 == 0) {
                if (few) {
                    hipLaunchKernelGGL((egrad_fixup_wg<4>), dim3(V), blk, 0, st, m->run_start, m->run_end, V, de,
                                       m->ehead, m->etail, m->g_re);
                } else {
                    hipLaunchKernelGGL((egrad_fixup<4>), fgrid, blk, 0, st, m->run_start, m->run_end, V, de,
                                       m->ehead, m->etail, m->g_re);
                }
            } else if (few) {
                hipLaunchKernelGGL((egrad_fixup_wg<1>), dim3(V), blk, 0, st, m->run_start, m->run_end, V, de,
                                   m->ehead, m->etail, m->g_re);
            } else {
                hipLaunchKernelGGL((egrad_fixup<1>), fgrid, blk, 0, st, m->run_start, m->run_end, V, de,
                                   m->ehead, m->etail, m->g_re);
            }
        }
#undef SERT_EG_ARGS
        }   // sorted path
        return 0;
    };
    bool dense_bound = false;
    auto dh_gemm = [&]() -> int {
        {
            // dh = da.W^T
            ScopedTimer t(m, TG_GEMM_DX);
#ifdef SERT_VARIANTS
            const bool strip = gemm_strip_ok(B, dw, de, de, de, true, m->DA, m->W);
#else
            const bool strip = false;
#endif
            // (ev_dense below: the completion signal of this GEMM, not a barrier packet behind it)
            dense_bound = m->lazy_join && ext_events() && !strip && !fork_nce;
            if (dense_bound) set_stop_event(fork_late ? m->ev_fork : m->ev_dense);
#ifdef SERT_VARIANTS
            if (fused_bwd) {
                // dh, the per-workgroup partial slabs of dW and their column sums (db): one launch
                static const bool attr_set = hipFuncSetAttribute((const void*)vs_bwd_fused, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 (int)vs_bwd_fused_lds_bytes()) == hipSuccess;
                if (!attr_set) { set_stop_event(nullptr); SERT_FAIL("cannot reserve the LDS of vs_bwd_fused"); }
                BwdFusedArgs fa;
                fa.DA = m->DA; fa.H = m->H; fa.W = m->W; fa.DH = m->DH; fa.part = m->part; fa.B = B;
                fa.stride = (size_t)FB_D * FB_D + FB_D;
                SERT_LAUNCH(vs_bwd_fused, dim3(fused_grid), dim3(FB_THREADS), vs_bwd_fused_lds_bytes(), m->stream, fa);
            } else
            if (strip)
                launch_gemm_strip<true, EPI_STORE>(m->stream, m->DA, m->W, m->DH, nullptr, B, dw, de, de, de, dw);
            else
#endif
                launch_gemm<false, true, EPI_STORE>(m->stream, m->DA, m->W, m->DH, nullptr, B, dw, de, de,
                                                    de, dw);
            set_stop_event(nullptr);
        }
        // From here on the main stream has produced dW, db and the loss partials AND is done
        // READING W (the dh GEMM): the side stream may update the small tensors.
        if (m->lazy_join && !dense_bound && !fork_nce) SERT_HIP(hipEventRecord(fork_late ? m->ev_fork : m->ev_dense, m->stream));
        if (fork_late && !fork_nce && !is_dp(m) && !m->timing.enabled && m->epart && c.kind == SERT_KIND_VECTORSPACE &&
            c.num_negatives > 0 && m->neg_alt_step != m->step + 1) {
            // The NEXT step's negatives (Philox position = the step counter after this step's update) are drawn NOW,
            // while the side stream still idles in front of the fork -- beside this step's gather / projection / loss
            // kernels -- instead of at the end of the step between the entity chain and the R_e update, where the
            // 5 us launch stretched to 18 us beside the word table's Adam and sat on the path to the tail (round 4:
            // the side chain ended 2.5 us AFTER the main stream's Adam).  neg_alt is free: this step's own negatives
            // were swapped into `neg` at its start.  Ordered before the next step's loss kernel by this stream's
            // order and the end-of-step join (ev_small).
            const int64_t count = (int64_t)c.batch_size * c.num_negatives;
            hipLaunchKernelGGL(vs_sample_negatives, dim3(grid_for((count + 3) / 4)), dim3(256), 0, m->stream2, m->neg_alt, count,
                               (int64_t)m->rank * count, (uint32_t)c.num_entities, c.seed, (uint64_t)(m->step + 1) * 2);
            m->neg_alt_step = m->step + 1;
        }
        if (fork_late && !fork_nce) SERT_HIP(hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
        if (fork_late && dw_third_queue(m)) SERT_HIP(hipStreamWaitEvent(m->stream3, m->ev_fork, 0));
        return 0;
    };
    auto word_table_sum = [&]() -> int {
        {
            ScopedTimer t(m, TG_SCATTER);
            // dR_w[X[i,k],:] += dh[i,:] / n
            SERT_TRY(word_grad_segsum(m, ds, batch_index, m->DH, (float)n));
        }
        return allreduce_word_grad(m);
    };
    auto dense_grad = [&]() -> int {
        // dW = h^T.da (reduction over the batch: split-K, order-fixed combine);
        // db = sum_i da_i rides along as the column sums of the da operand.
        // Third stream: dW and dh are both 512-workgroup launches (2 waves per SIMD, too
        // few to hide their own latencies) -- side by side they fill each other's bubbles.
        hipStream_t sd = (m->timing.enabled || m->nstreams < 3) ? m->stream : m->stream3;
        // Single GPU, two streams: dW, db and the loss partials only feed the small-tensor
        // optimiser and the loss, both of which already sit behind the entity chain on the side
        // stream -- issued there (behind that chain) they leave the main stream with nothing but
        // the dependency chain loss -> dh -> segmented sum -> word-table optimiser.
        // (measured: 0.376 -> 0.386 ms at C2 -- off by default, SERT_DW_SIDE=1 to try it)
        static const bool dw_side = variant_knob("SERT_DW_SIDE") && atoi(variant_knob("SERT_DW_SIDE")) != 0;
        if ((dw_side || side_heavy) && m->lazy_join) sd = m->stream2;
        if (m->dp_late_join || m->dw_side_first) sd = m->stream2;
        if (fork_late && m->lazy_join && dw_third_queue(m)) sd = m->stream3;
        if (sd != m->stream && sd != m->stream2 && !fork_late) SERT_HIP(hipStreamWaitEvent(sd, m->ev_fork, 0));
        // ~1024 workgroup items in all, at most 512 slabs (the optimum at one output tile: 512 slabs
        // of 128 rows) and at least 64 rows per slab.  With nine output tiles (d = 300) that is 114
        // slabs at batch >= 16384 and 64 at 4096 -- 512 / 256 slabs made the combine read up to 92 MB
        // of partials (sweep in DESIGN.md section 7.5).
        static const int user_splits = [] { const char* e = variant_knob("SERT_DW_SPLITS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
        int auto_splits = std::max(1, std::min(std::min(512, cdiv(1024, cdiv(dw, GM) * cdiv(de, GN))), B / 64));
        // the bf16-pipe kernel (gemm_x3.h) runs one workgroup per (k range, 160-column tile; one tile up to 128 x 128): one
        // workgroup per CU -- 256 slabs at C2 (0.2745 -> 0.2697 ms against 512; 128: 0.285), 128 at C4 (1.595 -> 1.579 ms)
        const int x3_splits = std::max(1, std::min(256 / ((dw <= 128 && de <= 128) ? 1 : cdiv(de, 160)), B / 64));
        if (gemm_x3_enabled() && x3_shape_ok(true, false, m->H, m->DA, dw, de, B, dw, de, x3_splits))
            auto_splits = x3_splits;
        const int want_splits = user_splits ? user_splits : auto_splits;
        int splits = std::min(want_splits, cdiv(B, GK));
        int kper = (int)round_up(cdiv(B, splits), GK);
        splits = cdiv(B, kper);
        const size_t mn = (size_t)dw * de;
        const size_t stride = mn + de;
        if (fused_bwd) {
            // (the partial slabs were written by vs_bwd_fused, behind which this runs)
            splits = fused_grid;
            if (sd != m->stream) SERT_FAIL("internal: the fused backward needs dW's combine on the main stream");
        } else
#ifdef SERT_VARIANTS
        if (gemm_strip_ok(B, de, dw, dw, de, false, m->H, m->DA) && dw % 32 == 0 && de % 4 == 0) {
            // strip kernel: every workgroup accumulates its contiguous strips' h^T.da (+ column sums)
            ScopedTimer t(m, TG_GEMM_DW);
            static const int want_wgs = variant_knob("SERT_STRIP_DW_WGS") ? atoi(variant_knob("SERT_STRIP_DW_WGS")) : 512;   // tuning knob
            const int strips = cdiv(B, SG_ROWS);
            const int spw = std::max(1, cdiv(strips, std::min(want_wgs, 1024)));
            splits = cdiv(strips, spw);
            hipLaunchKernelGGL(gemm_strip_tn, dim3(splits), dim3(256), 0, sd, (const float*)m->H, (const float*)m->DA, B,
                               dw, de, dw, de, spw, m->part, stride);
        } else
#endif
        {
            ScopedTimer t(m, TG_GEMM_DW);
            launch_gemm<true, false, EPI_STORE, true>(sd, m->H, m->DA, m->part, nullptr, dw, de,
                                                      B, dw, de, de, splits, kper, stride);
        }
        // single GPU: the combine rides in the step's tail launch (vs_tail) with the W, b update and
        // the loss finalisation
        static const bool no_tail = variant_knob("SERT_NO_TAIL") != nullptr;   // cross-check knob
        m->tail_splits = 0;
        // (side_heavy: the partial slabs come from the side stream, which is joined in front of the tail)
        if (!no_tail && !is_dp(m) && (sd == m->stream || ((side_heavy || m->dw_side_first) && sd == m->stream2)) &&
            m->cfg.kind == SERT_KIND_VECTORSPACE && !m->pt_big[2] &&
            mn + de < ((size_t)1 << 31)) {
            m->tail_splits = splits;
            m->tail_stride = stride;
            m->tail_part = m->part;
            // Experiment (round 6 item 14, variants build: SERT_COMBINE_SIDE=1): with dW / db first on the SIDE stream their split-K combine
            // there too, right behind the GEMM (the tail's own summation order: reduce_partials_g<16>), so that the tail reads 66 kB of sums
            // instead of the slabs.  Bit-identical and SLOWER at every batch size (C2 0.2294-0.2304 against 0.2226-0.2229 ms): the tail is a
            // latency-bound launch whatever it reads, and the combine lengthens the side chain.
            static const bool combine_side_off = !(variant_knob("SERT_COMBINE_SIDE") && atoi(variant_knob("SERT_COMBINE_SIDE")) == 1);
            if (!combine_side_off && m->dw_side_first && sd == m->stream2 && splits > 1 && !c.keep_grads) {
                ScopedTimer t(m, TG_SPLITK, sd);
                const size_t count = stride;
                hipLaunchKernelGGL((reduce_partials_g<16>), dim3((unsigned)((count + 63) / 64)), dim3(1024), 0, sd, (const float*)m->part, splits, stride,
                                   count, m->g_w, mn, m->g_b, (const int32_t*)nullptr, 0);
                m->tail_splits = 1;
                m->tail_part = m->g_w;         // (g_w | g_b are adjacent in the flat gradient buffer: one "slab" of mn + de sums)
                if (m->g_b != m->g_w + mn) SERT_FAIL("internal: g_W and g_b are not adjacent");
            }
        } else {
            ScopedTimer t(m, TG_SPLITK);
            launch_reduce_partials(sd, m->part,
                               splits, stride, stride, m->g_w, mn, m->g_b);
        }
        // the loss partials only depend on the NCE kernel too
        SERT_TRY(reduce_rowloss(m, sd));
        // timing knock-out (variants build, WRONG loss; r06 experiments item 9): the dW combine + W, b update right behind dW on the side
        // stream, the loss left to a one-workgroup launch behind the word table's update -- what splitting the tail that way would buy
        static const bool ko_tail_early = variant_knob("SERT_KO_TAIL_EARLY") != nullptr;
        m->tail_early = false;
        if (ko_tail_early && m->tail_splits > 0 && m->dw_side_first && sd == m->stream2 && !c.keep_grads && !m->timing.enabled) {
            AdamArgs aa2; AdadeltaArgs da2;
            optimizer_args(m, m->step + 1, &aa2, &da2);
            TailArgs ta;
            ta.part = m->tail_part ? m->tail_part : m->part; ta.splits = m->tail_splits; ta.stride = m->tail_stride;
            ta.W = m->W; ta.b = m->b; ta.s0_w = m->s0_w; ta.s1_w = m->s1_w; ta.s0_b = m->s0_b; ta.s1_b = m->s1_b;
            ta.g_w = m->g_w; ta.g_b = m->g_b;
            ta.n_w = (unsigned)m->n_w; ta.n_b = (unsigned)m->n_b;
            ta.aa = aa2;
            ta.loss_partials = m->red_loss; ta.n_loss = 0;
            ta.sq_partials = m->red_sq; ta.n_sq = 0;
            ta.sq_alt = nullptr; ta.sq_alt_lo = 0; ta.sq_alt_hi = 0;
            ta.inv_batch = 1.f; ta.reg_scale = 0.f;
            ta.out = m->d_loss; ta.host_flag = nullptr; ta.seq = 0u;
            ta.blk = m->tail_blk;
            if (++m->tail_launch_seq == 0) ++m->tail_launch_seq;
            ta.launch_seq = m->tail_launch_seq;
            const int nbt = cdiv((int64_t)(m->n_w + m->n_b), 64);
            hipLaunchKernelGGL((vs_tail<false>), dim3(nbt), dim3(1024), 0, sd, ta);
            m->tail_early = true;
        }
        if (m->dw_side_first) SERT_HIP(hipEventRecord(m->ev_dense, sd));   // (the tail waits for this, not for the chain behind it)
        if (sd != m->stream && sd != m->stream2 && !fork_late) SERT_HIP(hipEventRecord(m->ev_join3, sd));
        return 0;
    };
    // On a single GPU the only consumer of dR_e is the small-tensor optimiser, which runs on
    // the side stream right behind the entity chain: the main stream then never waits for
    // that chain, and the word-table optimiser starts straight after segsum instead of
    // idling ~12 us on a cross-queue dependency.
    m->lazy_join = !is_dp(m) && !m->timing.enabled && m->nstreams == 2 && (m->n_re <= ((size_t)1 << 22) || side_heavy);
    m->side_heavy = side_heavy;
    // Data parallel over an asynchronous communicator: nothing on the main stream needs what the side stream produces
    // (dR_e, and -- issued there too -- dW, db and the loss sum) before the all-reduce of the replicated remainder, and
    // that runs on the communication stream.  So the communication stream joins the side stream (allreduce_rest), the main
    // stream goes from the segmented sum straight to the hand-over of the word rows and their update: 40 us of dW GEMM,
    // combine and loss sum leave the critical path (C2, world of one: 0.329 -> 0.29 ms)
    // Single GPU, late fork: dW, db (and the loss partials) only feed the tail.  FIRST on the side stream -- in front of the
    // entity chain, beside the segmented sum -- they leave the main stream's dependency chain (loss -> dh -> segmented sum
    // -> word-table update -> tail) 20 us shorter; the tail waits for their event, which is long complete by then.
    static const bool dw_first_off = variant_knob("SERT_DW_FIRST") && atoi(variant_knob("SERT_DW_FIRST")) == 0;
    // Measured (tools/experiments/r04_dw_first*.sh, C2 dims): batch 4096 0.1176 -> 0.1092 ms, 8192 0.130 -> 0.116, 16384 0.1566 ->
    // 0.1429, 32768 0.191 -> 0.175; at 65536 0.2720 -> 0.2745 -- there dW streams its 67 MB beside the first level of the
    // segmented sum, whose 33.5 MB of dh rows then no longer stay in the Infinity Cache.  Taken while dh is below 24 MB.
    static const bool dw_first_always = variant_knob("SERT_DW_FIRST") && atoi(variant_knob("SERT_DW_FIRST")) == 2;
    // (!pt_big[2]: a projection matrix large enough for a streaming update of its own is updated on the main stream, which
    //  would then have to wait for the side stream's dW)
    const bool fork_nce_dw = fork_nce && fork_at_nce_dw(m) && !side_heavy && m->lazy_join && !fused_bwd && !m->pt_big[2] && m->nstreams == 2;
    // Round 6: the entity keys' partition goes out first of all on the side stream, in front of the fork wait (early_bucket above) --
    // and where it does, dW / db first on the side stream pays at EVERY batch size: the chain behind the fork is then dW + egrad_acc,
    // the main stream goes from dh straight into the tree.  tools/experiments/r06_dw_first_again.sh, three rounds on one box, ms/step,
    // dW on the main stream / first on the side stream: batch 65536 0.2375-0.2390 / 0.2244-0.2261 (-5.4 %; with the partition behind the
    // fork, as in round 5: 0.2381-0.2404 / 0.2346-0.2360), 131072 0.4190-0.4222 / 0.4040-0.4142.
    SERT_TRY(early_bucket());
    SERT_TRY(early_sort());
    static const bool chain_behind_tree = variant_knob("SERT_CHAIN_BEHIND_TREE") != nullptr;
    m->dw_side_first = fork_nce_dw ||
                       (!dw_first_off && fork_late && !fork_nce && !side_heavy && m->lazy_join && !fused_bwd && !dw_third_queue(m) &&
                        !m->pt_big[2] && c.kind == SERT_KIND_VECTORSPACE &&
                        (dw_first_always || (((size_t)B * dw * sizeof(float) <= ((size_t)24 << 20) || m->bucket_early) && m->epart)));
    // (m->epart: the sort-free entity chain of small entity tables.  Behind the counting sort of a larger one the side stream is
    //  the longer of the two already: the reference's product-search settings, V_e = 32768, 205.8 -> 214.5 us with dW in front)
    static const int dp_late_mode = variant_knob("SERT_DP_LATE") ? atoi(variant_knob("SERT_DP_LATE")) : 1;   // 0: off; 2: dW behind the chain
    const bool dp_late = dp_late_mode != 0 && is_dp(m) && !m->host_ar && m->comm && !m->timing.enabled && m->nstreams == 2 && !side_heavy && !fork_nce &&
                         !fork_late;
    m->dp_late_join = dp_late;
    if (side_heavy && chain_behind_tree && m->sort_early) {
        // (experiment, round 6 item 13: with the key sort beside the forward, the rest of the sorted entity chain -- chunked reduce, fix-up, then
        //  dW -- forked behind the word gradient's TREE instead of behind the loss kernel: beside the update, not beside the tree)
        SERT_TRY(dh_gemm());
        SERT_TRY(word_table_sum());
        m->fork_bound = false;         // (not the loss kernel's completion signal: a record behind the tree)
        SERT_TRY(entity_grad());       // (records its fork on the main stream HERE: behind the tree)
        SERT_TRY(dense_grad());
    } else if (side_heavy) {
        SERT_TRY(entity_grad());       // side, forked on the loss kernel's completion
        SERT_TRY(dh_gemm());           // main (its completion is ev_dense)
        SERT_TRY(word_table_sum());    // main
        SERT_TRY(dense_grad());        // side, behind the entity chain
    } else if (fork_nce && fork_nce_dw) {
        if (!m->fork_bound) SERT_HIP(hipEventRecord(m->ev_fork, m->stream));
        m->fork_bound = false;
        SERT_HIP(hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
        SERT_TRY(dh_gemm());           // main
        SERT_TRY(dense_grad());        // side, beside the dh GEMM
        SERT_TRY(entity_grad());       // side, behind dW
        SERT_TRY(word_table_sum());    // main
    } else if (fork_nce) {
        SERT_TRY(entity_grad());       // side, forked on the NCE kernel's completion
        SERT_TRY(dh_gemm());
        SERT_TRY(dense_grad());
        SERT_TRY(word_table_sum());
    } else if (fork_late) {
        SERT_TRY(dh_gemm());           // main; its completion is the step's one fork
        if (m->dw_side_first) SERT_TRY(dense_grad());   // side, in front of the entity chain
        SERT_TRY(entity_grad());       // side
        if (!m->dw_side_first) SERT_TRY(dense_grad());  // main (W and b are then updated on the main stream too)
        SERT_TRY(word_table_sum());    // main
    } else if (is_dp(m)) {
        // data parallel: the word-table gradient first, so that its exchange (rows' all-to-all or
        // reduce-scatter) overlaps dW and the entity chain (dW in front of the segmented sum instead:
        // 0.362 -> 0.370 ms with a world of one -- the hand-over then sits bare on the critical path)
        if (dp_late && dp_late_mode == 2) {
            SERT_TRY(entity_grad());
            SERT_TRY(dh_gemm());
            SERT_TRY(word_table_sum());
            SERT_TRY(dense_grad());        // (side, behind the entity chain)
        } else if (dp_late) {
            // the side stream takes dW, db and the loss sum FIRST (beside dh and the segmented sum), then the entity chain:
            // behind that chain they ran beside the word table's Adam, three times as long, and the small all-reduce --
            // which waits for them -- ended 35 us after the Adam (0.330 ms; this order: 0.29)
            if (!m->fork_bound) SERT_HIP(hipEventRecord(m->ev_fork, m->stream));   // (else: the loss kernel's own completion signal)
            SERT_HIP(hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
            // The data-parallel step is bound by the HOST (some 45 runtime calls + three collectives per step: round-5 API
            // trace, tools/experiments/r05_hip_trace.sh with SERT_FORCE_COMM=1): the launches go out in order of
            // criticality -- the main stream's dh GEMM first; issued behind the six side-stream launches it started 27 us
            // after the loss kernel had finished (C2, world of one).
            static const bool dx_last = variant_knob("SERT_DP_DX_LAST") != nullptr;   // (the round-4 order, for the A/B)
            if (!dx_last) SERT_TRY(dh_gemm());
            SERT_TRY(dense_grad());
            m->fork_bound = true;          // (the fork is recorded: the entity chain only has to follow in stream order)
            SERT_TRY(entity_grad());
            if (dx_last) SERT_TRY(dh_gemm());
            SERT_TRY(word_table_sum());
        } else {
            SERT_TRY(entity_grad());
            SERT_TRY(dh_gemm());
            SERT_TRY(word_table_sum());
            SERT_TRY(dense_grad());
        }
    } else if (fused_bwd) {
        SERT_TRY(entity_grad());
        SERT_TRY(dh_gemm());           // (dh and the dW partials in one launch)
        SERT_TRY(dense_grad());
        SERT_TRY(word_table_sum());
    } else {
        // single GPU: the MFMA-bound dW beside the latency-bound sort of the side stream
        SERT_TRY(entity_grad());
        SERT_TRY(dense_grad());
        SERT_TRY(dh_gemm());           // (records ev_dense behind the dX GEMM)
        SERT_TRY(word_table_sum());
    }
    // join the entity-gradient chain (and the dense gradients of a third stream)
    if (!m->lazy_join && !m->dp_late_join && !m->timing.enabled && m->nstreams >= 2) {
        SERT_HIP(hipEventRecord(m->ev_join, m->stream2));
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    }
    if (!m->timing.enabled && m->nstreams >= 3) SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_join3, 0));
    (void)row0;
    return 0;
}

// ---- the full-softmax vectorspace variant (additive) ----------------------------
template <bool TRAIN>
static int fs_forward(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, dw = c.word_dim, de = c.entity_dim, V = c.num_entities;
    const size_t row0 = (size_t)batch_index * B;
    {
        ScopedTimer t(m, TG_GATHER);
        SERT_ID_DISPATCH(c.id_bytes, {
            const IdT* X = (const IdT*)ds.x + row0 * n;
            if (dw % 4 == 0)
                hipLaunchKernelGGL((vs_gather_mean<IdT, 4>), dim3(grid_for((int64_t)B * dw / 4, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, X, m->rw, m->H, B, n, dw);
            else
                hipLaunchKernelGGL((vs_gather_mean<IdT, 1>), dim3(grid_for((int64_t)B * dw, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, X, m->rw, m->H, B, n, dw);
        });
    }
    {
        ScopedTimer t(m, TG_GEMM_FWD);
        launch_gemm<false, false, EPI_BIAS_TANH>(m->stream, m->H, m->W, m->T, m->b, B, de, dw, dw, de, de);
        // p = clip(t) ; logits = p.R_e^T   (B, V)
        hipLaunchKernelGGL(vs_clip, dim3(grid_for((int64_t)B * de)), dim3(256), 0, m->stream, m->T, m->DH2,
                           (size_t)B * de);
    }
    const float inv_batch = 1.0f / (float)c.global_batch_size;
    const int tile = m->fs_tile > 0 ? m->fs_tile : B;
    for (int r0 = 0; r0 < B; r0 += tile) {
        const int rows = std::min(tile, B - r0);
        {
            ScopedTimer t(m, TG_GEMM_FWD);
            launch_gemm<false, true, EPI_STORE>(m->stream, m->DH2 + (size_t)r0 * de, m->re, m->Z, nullptr, rows, V, de, de, de, V);
        }
        {
            ScopedTimer t(m, TG_LOSS);
            // rows up to 2048 entities stay in registers (one read, one write)
#define SERT_FS_CE(EPL)                                                                                   \
    hipLaunchKernelGGL((fs_softmax_ce<TRAIN, EPL>), dim3(cdiv(rows, 4)), dim3(256), 0, m->stream, m->Z, \
                       ds.y + row0 + r0, TRAIN ? ds.w + row0 + r0 : nullptr, m->rowloss + r0, rows, V, inv_batch)
            if (V <= 64 * 16)      SERT_FS_CE(16);
            else if (V <= 64 * 32) SERT_FS_CE(32);
            else                   SERT_FS_CE(0);
#undef SERT_FS_CE
        }
        if (TRAIN && tile < B) {
            // row tiles: this tile's share of the backward that needs its dZ, before the next tile's
            // logits overwrite it -- dR_e += dZ_t^T.p_t (the first tile stores), dp_t = dZ_t.R_e
            {
                ScopedTimer t(m, TG_EGRAD);
                if (r0 == 0)
                    launch_gemm<true, false, EPI_STORE>(m->stream, m->Z, m->DH2, m->g_re, nullptr, V, de, rows, V, de, de);
                else
                    launch_gemm<true, false, EPI_ACCUM>(m->stream, m->Z, m->DH2 + (size_t)r0 * de, m->g_re, nullptr, V, de, rows, V, de, de);
            }
            {
                ScopedTimer t(m, TG_GEMM_DX);
                SERT_TRY((gemm_long_k<false, false>(m, m->stream, m->Z, m->re, m->DA + (size_t)r0 * de, rows, de, V, V, de)));
            }
        }
    }
    return 0;
}

static int fs_backward(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, dw = c.word_dim, de = c.entity_dim, V = c.num_entities;
    const bool tiled = m->fs_tile > 0 && m->fs_tile < B;   // (fs_forward already consumed every tile's dZ)
    if (!tiled) {
        // dR_e (V, d_e) = dZ^T.p : reduction over the batch, split-K, order-fixed combine
        ScopedTimer t(m, TG_EGRAD);
        const int tiles = cdiv(de, GN) * cdiv(V, GM);
        int splits = std::max(1, std::min(cdiv(B, GK), cdiv(1024, tiles)));
        int kper = (int)round_up(cdiv(B, splits), GK);
        splits = cdiv(B, kper);
        const size_t mn = (size_t)V * de;
        launch_gemm<true, false, EPI_STORE>(m->stream, m->Z, m->DH2, m->part, nullptr, V, de, B, V, de, de,
                                            splits, kper, mn);
        launch_reduce_partials(m->stream, m->part, splits, mn,
                           mn, m->g_re, mn, m->g_re);
    }
    {
        // dp = dZ.R_e (B, d_e) ; da = dp * clip'(t) * tanh'(a)
        ScopedTimer t(m, TG_GEMM_DX);
        if (!tiled) SERT_TRY((gemm_long_k<false, false>(m, m->stream, m->Z, m->re, m->DA, B, de, V, V, de)));
        hipLaunchKernelGGL(vs_tanh_backward, dim3(grid_for((int64_t)B * de)), dim3(256), 0, m->stream, m->DA,
                           m->T, (size_t)B * de);
    }
    {
        static const int want_splits = [] { const char* e = variant_knob("SERT_DW_SPLITS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
        int splits = std::min(want_splits, cdiv(B, GK));
        int kper = (int)round_up(cdiv(B, splits), GK);
        splits = cdiv(B, kper);
        const size_t mn = (size_t)dw * de;
        const size_t stride = mn + de;
        {
            ScopedTimer t(m, TG_GEMM_DW);
            launch_gemm<true, false, EPI_STORE, true>(m->stream, m->H, m->DA, m->part, nullptr, dw, de, B, dw, de,
                                                      de, splits, kper, stride);
        }
        {
            ScopedTimer t(m, TG_SPLITK);
            launch_reduce_partials(m->stream, m->part, splits,
                               stride, stride, m->g_w, mn, m->g_b);
        }
        launch_gemm<false, true, EPI_STORE>(m->stream, m->DA, m->W, m->DH, nullptr, B, dw, de, de, de, dw);
    }
    {
        ScopedTimer t(m, TG_SCATTER);
        SERT_TRY(word_grad_segsum(m, ds, batch_index, m->DH, (float)n));
    }
    SERT_TRY(allreduce_word_grad(m));
    return 0;
}

// ---- the loglinear step -------------------------------------------------------
// Streaming loss for entity vocabularies beyond the LDS-resident slab (kernels_ll.h).
template <bool TRAIN, bool V4>
static int ll_stream_loss(sert_model* m, const DataSplit& ds, size_t row0, const int32_t* y,
                          const int64_t* indptr, const float* w, float inv_batch, const int32_t* slot) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, V = c.num_entities;
    const int64_t rows = (int64_t)B * n;
    const int nseg = cdiv(V, kLlSeg);
    hipStream_t s = m->stream;
    if (TRAIN && !ds.labfix) SERT_FAIL("training split has no label scratch");
    float* labfix = TRAIN ? ds.labfix + (y ? row0 : 0) : nullptr;
    // slot: the logits live in the distinct-word table Zu (U rows); the per-token log-sum-exp
    // is then a per-WORD quantity, and dL/dZ of every token goes to Z
    const float* logits = slot ? m->Zu : m->Z;
    const int64_t lrows = slot ? m->ll_U : rows;
    hipLaunchKernelGGL((ll_s_tokstat<V4>), dim3((unsigned)(lrows * nseg)), dim3(256), 0, s, logits, V, nseg, m->ll_tokstat);
    hipLaunchKernelGGL(ll_s_lse, dim3(cdiv(lrows, 4)), dim3(256), 0, s, m->ll_tokstat, lrows, nseg, m->ll_lse);
    hipLaunchKernelGGL((ll_s_window<V4>), dim3((unsigned)((int64_t)B * nseg)), dim3(256), 0, s, logits, m->ll_lse, n, V,
                       nseg, m->J, m->ll_jstat, slot);
    hipLaunchKernelGGL((ll_s_rowloss<TRAIN>), dim3(B), dim3(256), 0, s, m->J, m->ll_jstat, y, indptr,
                       ds.csr_indices, ds.csr_data, w, m->rowloss, m->ll_rowinfo, labfix, V, nseg, inv_batch);
    if (!TRAIN) return 0;
    hipLaunchKernelGGL((ll_s_dj<V4>), dim3((unsigned)((int64_t)B * nseg)), dim3(256), 0, s, m->J, m->ll_rowinfo, V, nseg);
    hipLaunchKernelGGL(ll_s_labfix, dim3(B), dim3(256), 0, s, m->J, y, indptr, ds.csr_indices, labfix, V);
    hipLaunchKernelGGL((ll_s_tokr<V4>), dim3((unsigned)(rows * nseg)), dim3(256), 0, s, logits, m->ll_lse, m->J, n, V,
                       nseg, m->ll_rpart, slot);
    hipLaunchKernelGGL(ll_s_rsum, dim3(cdiv(rows, 4)), dim3(256), 0, s, m->ll_rpart, rows, nseg, m->ll_r);
    if (slot) {
        // distinct-word mode: stop here -- dJ (in J) and r_ik are all the per-word backward
        // needs (dzu_from_dj); the per-token dL/dZ pass and its 2 x B*n*V_e floats are skipped.
        // The word rows must hold LOG-probabilities for the finishing transform:
        hipLaunchKernelGGL(ll_s_logp_rows, dim3((unsigned)(lrows * nseg)), dim3(256), 0, s, m->Zu, m->ll_lse, V, nseg);
        return 0;
    }
    hipLaunchKernelGGL((ll_s_dz<V4>), dim3((unsigned)(rows * nseg)), dim3(256), 0, s, m->Z, m->ll_lse, m->J, m->ll_r, n,
                       V, nseg);
    return 0;
}

template <bool TRAIN>
static int ll_forward(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, d = c.word_dim, V = c.num_entities;
    const size_t row0 = (size_t)batch_index * B;
    const int64_t rows = (int64_t)B * n;
    const size_t fused_lds = ((size_t)n * V + V) * sizeof(float);
    const bool fused = fused_lds <= 150 * 1024 && ds.max_labels_per_row <= 1024;
    // Duplicate tokens share their logit row (Z[r,:] = R_w[X[r],:].W + b depends on the word
    // only): in a training step the gather and all three GEMMs run on the batch's DISTINCT
    // words (Zipfian batches: a third of the tokens), the loss kernel reads the table through
    // the per-token slot, and dL/dZ is summed per word before the backward GEMMs.
    static const bool no_dedup = knob("SERT_LL_NODEDUP") != nullptr;   // cross-check knob
    static const bool rowwise = variant_knob("SERT_LL_ROWWISE") != nullptr;
    // (evaluation passes over the TRAINING split -- train_error() -- have the index too)
    static const bool eval_dedup = variant_knob("SERT_LL_NO_EVAL_DEDUP") == nullptr;
    m->ll_dedup = (TRAIN || eval_dedup) && (fused || !rowwise) && !no_dedup && ds.idx_slots != nullptr &&
                  (size_t)batch_index < ds.idx_batches.size();
    const BatchIndex* bx = m->ll_dedup ? &ds.idx_batches[(size_t)batch_index] : nullptr;
    m->ll_U = bx ? bx->num_distinct : 0;
    const int64_t grows = m->ll_dedup ? m->ll_U : rows;          // rows of the gathered operand
    {
        ScopedTimer t(m, TG_GATHER);
        if (m->ll_dedup) {
            const uint32_t* U = reinterpret_cast<const uint32_t*>(ds.idx_uwords + bx->uw_off);
            if (d % 4 == 0)
                hipLaunchKernelGGL((ll_gather_rows<uint32_t, 4>), dim3(grid_for(grows * d / 4, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, U, m->rw, m->G, grows, d);
            else
                hipLaunchKernelGGL((ll_gather_rows<uint32_t, 1>), dim3(grid_for(grows * d, 256, 1 << 20)),
                                   dim3(256), 0, m->stream, U, m->rw, m->G, grows, d);
        } else {
            SERT_ID_DISPATCH(c.id_bytes, {
                const IdT* X = (const IdT*)ds.x + row0 * n;
                if (d % 4 == 0)
                    hipLaunchKernelGGL((ll_gather_rows<IdT, 4>), dim3(grid_for(rows * d / 4, 256, 1 << 20)),
                                       dim3(256), 0, m->stream, X, m->rw, m->G, rows, d);
                else
                    hipLaunchKernelGGL((ll_gather_rows<IdT, 1>), dim3(grid_for(rows * d, 256, 1 << 20)),
                                       dim3(256), 0, m->stream, X, m->rw, m->G, rows, d);
            });
        }
    }
    {
        ScopedTimer t(m, TG_GEMM_FWD);
        launch_gemm<false, false, EPI_BIAS>(m->stream, m->G, m->W, m->ll_dedup ? m->Zu : m->Z, m->b, (int)grows,
                                            V, d, d, V, V);
    }
    const float inv_batch = 1.0f / (float)c.global_batch_size;
    const int32_t* y = ds.y ? ds.y + row0 : nullptr;
    const int64_t* indptr = ds.csr_indptr ? ds.csr_indptr + row0 : nullptr;
    const float* w = TRAIN ? ds.w + row0 : nullptr;
    const int32_t* slot = m->ll_dedup ? ds.idx_slots + (size_t)batch_index * rows : nullptr;
    // fused path: the row's (n, V) slab lives in LDS; CSR rows with > 1024 labels fall back
    if (fused) {
        ScopedTimer t(m, TG_LOSS);
        if (m->ll_dedup)   // the per-token log-softmax, once per distinct word
        {
            if (V <= 64 * 16)      hipLaunchKernelGGL((ll_logsoftmax_rows<16>), dim3(cdiv(grows, 4)), dim3(256), 0, m->stream, m->Zu, grows, V);
            else if (V <= 64 * 32) hipLaunchKernelGGL((ll_logsoftmax_rows<32>), dim3(cdiv(grows, 4)), dim3(256), 0, m->stream, m->Zu, grows, V);
            else                   hipLaunchKernelGGL((ll_logsoftmax_rows<0>), dim3(cdiv(grows, 4)), dim3(256), 0, m->stream, m->Zu, grows, V);
        }
        // 512 threads per row: 302 us at 256 (too few waves to hide the slab load), 228 at
        // 512, 320 at 640 (one wave per token, but only two workgroups fit a CU)
        // (distinct-word mode: the kernel writes dJ_i into J and r_ik into ll_r instead of dL/dZ)
        static const bool slab = variant_knob("SERT_LL_SLAB") != nullptr;   // cross-check knob
        if (TRAIN && m->ll_dedup && n <= 64 && !slab) {
            // distinct-word mode: no LDS slab, the n table rows are read once, coalesced along e
            const size_t lds = ((size_t)V + n) * sizeof(float);
            // a row is latency-bound (a handful of barriers), not work-bound: 128-thread workgroups
            // put four times as many rows on a CU -- loss kernel 307 -> 111 us at batch 65536, V_e = 100;
            // 86 -> 75 us at V_e = 1000, 139 -> 134 us at 2000 (batch 8192)
            static const int nt128_below = variant_knob("SERT_LL_NT128_BELOW") ? atoi(variant_knob("SERT_LL_NT128_BELOW")) : 2048;   // tuning knob
            // up to 2048 entities (V_e % 4 == 0): one WAVE per row, the row in registers, no LDS and no barrier
            static const bool no_wave = variant_knob("SERT_LL_NO_ROW_WAVE") != nullptr;   // cross-check knob
#define SERT_LL_WAVE(E)                                                                                        \
    hipLaunchKernelGGL((ll_row_wave<E>), dim3(cdiv(B, 4)), dim3(256), 0, m->stream, (const float*)m->Zu, slot, y, \
                       indptr, ds.csr_indices, ds.csr_data, w, m->rowloss, B, n, V, inv_batch, m->J, m->ll_r)
            // (its row fetches are buffer loads off one descriptor of the table: 32-bit byte offsets)
            if (!no_wave && V % 4 == 0 && V <= 2048 && (size_t)m->ll_U * V * sizeof(float) < ((size_t)1 << 32)) {
                const int e4 = cdiv(V / 4, 64);
                if (e4 <= 1) SERT_LL_WAVE(1);
                else if (e4 <= 2) SERT_LL_WAVE(2);
                else if (e4 <= 4) SERT_LL_WAVE(4);
                else SERT_LL_WAVE(8);
            } else
#undef SERT_LL_WAVE
            if (V <= nt128_below)
                hipLaunchKernelGGL((ll_row_from_table<128>), dim3(B), dim3(128), lds, m->stream, (const float*)m->Zu,
                                   slot, y, indptr, ds.csr_indices, ds.csr_data, w, m->rowloss, n, V, inv_batch,
                                   m->J, m->ll_r);
            else
                hipLaunchKernelGGL((ll_row_from_table<512>), dim3(B), dim3(512), lds, m->stream, (const float*)m->Zu,
                                   slot, y, indptr, ds.csr_indices, ds.csr_data, w, m->rowloss, n, V, inv_batch,
                                   m->J, m->ll_r);
        } else {
            hipLaunchKernelGGL((ll_fused_row<TRAIN, 512>), dim3(B), dim3(512), fused_lds, m->stream,
                               m->ll_dedup ? m->J : m->Z, (const float*)m->Zu, slot, y, indptr, ds.csr_indices,
                               ds.csr_data, w, m->rowloss, n, V, inv_batch, m->ll_r);
        }
    } else if (rowwise) {
        // the plain row-per-workgroup kernels (kept as a cross-check of the streaming path)
        ScopedTimer t(m, TG_LOSS);
        hipLaunchKernelGGL(ll_softmax_rows, dim3(cdiv(rows, 4)), dim3(256), 0, m->stream, m->Z, rows,
                           V);
        hipLaunchKernelGGL((ll_window<TRAIN>), dim3(B), dim3(256), 0, m->stream, m->Z, m->J, y,
                           indptr, ds.csr_indices, ds.csr_data, w, m->rowloss, n, V, inv_batch);
    } else {
        ScopedTimer t(m, TG_LOSS);
        if (V % 4 == 0) SERT_TRY((ll_stream_loss<TRAIN, true>(m, ds, row0, y, indptr, w, inv_batch, slot)));
        else            SERT_TRY((ll_stream_loss<TRAIN, false>(m, ds, row0, y, indptr, w, inv_batch, slot)));
    }
    return 0;
}

static int ll_backward(sert_model* m, const DataSplit& ds, int64_t batch_index) {
    const auto& c = m->cfg;
    const int B = c.batch_size, n = c.window_size, d = c.word_dim, V = c.num_entities;
    const size_t row0 = (size_t)batch_index * B;
    int64_t rows = (int64_t)B * n;          // rows of the dZ operand of the two GEMMs
    const float* dZ = m->Z;
    bool dg_mapped = false;                 // dG went straight to the word-table gradient rows
    if (m->ll_dedup) {
        // per-word sums of dL/dZ (the backward of "duplicate tokens share a logit row")
        ScopedTimer t(m, TG_EGRAD);
        SERT_TRY(dzu_from_dj(m, ds, batch_index));   // (every distinct-word step emits dJ_i + r_ik)
        dZ = m->dZu;
        rows = m->ll_U;
    }
    {
        // dW (d, V) = G^T.dZ, reduction over the tokens (distinct words); db = column sums of dZ
        const int tiles = cdiv(V, GN) * cdiv(d, GM);
        static const int want_items = variant_knob("SERT_LL_DW_ITEMS") ? std::max(1, atoi(variant_knob("SERT_LL_DW_ITEMS"))) : 1024;   // tuning knob
        int splits = std::max(1, std::min(cdiv(rows, GK), cdiv(want_items, tiles)));
        int kper = (int)round_up(cdiv(rows, splits), GK);
        splits = cdiv(rows, kper);
        const size_t mn = (size_t)d * V;
        const size_t stride = mn + V;
        // Single GPU, a GEMM worth forking for: dW, its combine and (optimizer_and_loss) the W, b update only
        // feed the loss finalisation -- they run on the side stream beside dG -> row scatter -> word-table
        // update instead of in front of them.  SERT_LL_DW_SIDE=0 keeps the whole step on one stream.
        static const bool dw_side_off = knob("SERT_LL_DW_SIDE") && atoi(knob("SERT_LL_DW_SIDE")) == 0;
        const bool dw_side = !dw_side_off && !is_dp(m) && !m->timing.enabled && m->nstreams >= 2 && ext_events() &&
                             2.0 * (double)rows * d * V >= 2e9;
        hipStream_t sd = dw_side ? m->stream2 : m->stream;
        m->ll_dw_side = dw_side;   // (optimizer_and_loss: W and b are updated on that stream too, whatever their size)
        if (dw_side) {
            SERT_HIP(hipEventRecord(m->ev_fork, m->stream));
            SERT_HIP(hipStreamWaitEvent(sd, m->ev_fork, 0));
        }
        {
            ScopedTimer t(m, TG_GEMM_DW);
            launch_gemm<true, false, EPI_STORE, true>(sd, m->G, dZ, m->part, nullptr, d, V,
                                                      (int)rows, d, V, V, splits, kper, stride);
        }
        {
            ScopedTimer t(m, TG_SPLITK);
            launch_reduce_partials(sd, m->part,
                               splits, stride, stride, m->g_w, mn, m->g_b);
        }
        {
            // dG (rows, d) = dZ.W^T -- in distinct-word mode row u IS the gradient of word uwords[u]: where the
            // 64x64-tile kernel takes the launch its epilogue stores the rows straight into dR_w (14.8 us of
            // scatter pass less on the step's chain at C2 dims); keep_grads keeps dG readable
            ScopedTimer t(m, TG_GEMM_DX);
            static const bool no_map = variant_knob("SERT_LL_NO_ROWMAP") != nullptr;   // cross-check knob
            const int32_t* rowmap = (m->ll_dedup && !c.keep_grads && !no_map)
                                        ? ds.idx_uwords + ds.idx_batches[(size_t)batch_index].uw_off : nullptr;
            SERT_TRY((gemm_long_k<false, true>(m, m->stream, dZ, m->W, m->DG, (int)rows, d, V, V, V, rowmap, m->g_rw, &dg_mapped)));
        }
    }
    if (!dg_mapped) {
        ScopedTimer t(m, TG_SCATTER);
        if (m->ll_dedup) {
            // dG already holds one row per distinct word: dR_w[word_u, :] = dG[u, :]
            const BatchIndex& bx = ds.idx_batches[(size_t)batch_index];
            hipLaunchKernelGGL(ll_scatter_rows, dim3(grid_for(rows * d)), dim3(256), 0, m->stream, m->DG,
                               ds.idx_uwords + bx.uw_off, rows, d, m->g_rw,
                               (unsigned char*)nullptr);
        } else {
            // dR_w[X[r],:] += dG[r,:]
            SERT_TRY(word_grad_segsum(m, ds, batch_index, m->DG, 1.0f));
        }
    }
    SERT_TRY(allreduce_word_grad(m));
    (void)row0;
    return 0;
}

// ---- shared tail: loss sum, exchange, optimiser, loss ---------------------------
// rowloss (B) -> per-block partials in red_loss; returns their count.  In a
// data-parallel step the partials are folded into the scalar slot of the flat
// gradient buffer so that the loss sum rides in the all-reduce.
static int reduce_rowloss(sert_model* m, hipStream_t st) {
    const int B = m->cfg.batch_size;
    int nb = std::min(kOptBlocks, cdiv(B, 256));
    m->loss_from_rows = false;
    if (is_vs(m) && !is_fs(m) && m->nce_loss_partials > 0) nb = m->nce_loss_partials;   // written by vs_nce
    else if (!is_dp(m) && B <= 8192) {
        // few rows: the finalisation sums them itself (fp64, fixed order) -- one 4 us launch less on a small step's chain
        m->loss_from_rows = true;
        nb = B;
    } else hipLaunchKernelGGL(sum_partial, dim3(nb), dim3(256), 0, st, m->rowloss, (size_t)B, m->red_loss);
    if (is_dp(m))
        hipLaunchKernelGGL(partials_to_scalar, dim3(1), dim3(256), 0, st, m->red_loss, nb, m->g_loss);
    m->n_loss_partials = nb;
    return 0;
}

// One streaming optimiser launch over `count` elements (kernels_opt.h).
static void launch_stream_opt(sert_model* m, hipStream_t st, float* p, float* g, float* s0, float* s1, size_t count,
                              int nb, const AdamArgs& aa, const AdadeltaArgs& da, float* sq,
                              const uint32_t* bits, unsigned row_len, int rows_mode = kRowsAll, float* sq_new = nullptr) {
    const bool keep = m->cfg.keep_grads != 0;
    if (is_vs(m)) {
        if (keep) hipLaunchKernelGGL((adam_l2<true>), dim3(nb), dim3(256), 0, st, p, g, s0, s1, count, aa, sq, bits, row_len, rows_mode, sq_new);
        else      hipLaunchKernelGGL((adam_l2<false>), dim3(nb), dim3(256), 0, st, p, g, s0, s1, count, aa, sq, bits, row_len, rows_mode, sq_new);
    } else {
        if (keep) hipLaunchKernelGGL((adadelta_l2<true>), dim3(nb), dim3(256), 0, st, p, g, s0, s1, count, da, sq, bits, row_len, rows_mode);
        else      hipLaunchKernelGGL((adadelta_l2<false>), dim3(nb), dim3(256), 0, st, p, g, s0, s1, count, da, sq, bits, row_len, rows_mode);
    }
}

// Optimiser hyper-parameters of optimiser step `t` (1-based; Adam's bias correction, models.py:922).
static void optimizer_args(const sert_model* m, int64_t t, AdamArgs* aa, AdadeltaArgs* da) {
    const auto& c = m->cfg;
    const float l2k = c.lambda_ > 0.f ? c.lambda_ / (float)c.global_batch_size : 0.f;
    *aa = AdamArgs{l2k, 0.f, c.beta1, c.beta2, c.eps};
    *da = AdadeltaArgs{l2k, c.lr, c.beta1, c.eps};
    if (is_vs(m)) {
        const float tf = (float)t;
        aa->a_t = c.lr * sqrtf(1.0f - powf(c.beta2, tf)) / (1.0f - powf(c.beta1, tf));
    }
}

// The word-table rows no token of this batch points to (their gradient is the L2 term alone)
// are not read by the batch's forward either: their update is issued NOW, at the start of the
// committed step, on its own stream, and runs beside forward and backward; optimizer_and_loss
// then only has the touched rows left on the critical path.
// Measured at C2 and C4 (profiles/r02b_variants.txt): SLOWER than one launch behind the backward
// (0.376 -> 0.391 ms at C2, 2.13 -> 2.25 ms at C4) -- the step is memory-system-bound from end to
// end, a second queue adds no bandwidth, and two row-filtered launches stream worse than one dense
// one.  Kept as an opt-in (SERT_ADAM_SPLIT=1) with its tests; off by default.
static bool adam_split_enabled() {
    static const bool on = variant_knob("SERT_ADAM_SPLIT") && atoi(variant_knob("SERT_ADAM_SPLIT")) != 0;
    return on;
}
static int issue_untouched_rows_update(sert_model* m, const uint32_t* bits) {
    m->early_issued = false;
    m->early_sq = 0;
    const bool split = adam_split_enabled();
    if (!split || !bits || !m->use_touched || is_dp(m) || m->timing.enabled || m->nstreams < 2) return 0;
    AdamArgs aa; AdadeltaArgs da;
    optimizer_args(m, m->step + 1, &aa, &da);
    // (the previous step's touched-row launch may have updated rows this launch owns)
    SERT_HIP(hipStreamWaitEvent(m->stream4, m->ev_word_opt, 0));
    const int nb = (int)std::min<int64_t>(kOptBlocks, cdiv(cdiv(m->n_rw, 4), 256));
    launch_stream_opt(m, m->stream4, m->rw, m->g_rw, m->s0_rw, m->s1_rw, m->n_rw, nb, aa, da, m->red_sq, bits,
                      (unsigned)m->cfg.word_dim, kRowsUntouched);
    SERT_HIP(hipEventRecord(m->ev_early, m->stream4));
    m->early_issued = true;
    m->early_sq = nb;
    return 0;
}

// loss_dst: device [3], or the pinned host block (publish = true: its sequence number is
// stored after the values, for the host to spin on)
static int optimizer_and_loss(sert_model* m, float* loss_dst, bool publish = false,
                              const uint32_t* bits = nullptr, const uint32_t* next_bits = nullptr) {
    const int n_loss_partials = m->n_loss_partials;
    const auto& c = m->cfg;
    const float l2k = c.lambda_ > 0.f ? c.lambda_ / (float)c.global_batch_size : 0.f;
    m->step += 1;
    AdamArgs aa; AdadeltaArgs da;
    optimizer_args(m, m->step, &aa, &da);
    // (partials [0, early_sq) belong to the untouched-row launch issued at the start of the step)
    int n_sq = m->early_issued ? m->early_sq : 0;
    const bool exchanged = m->comm && !m->timing.enabled;
    // single GPU: the small tensors are updated on the side stream WHILE the word table
    // streams on the main one (independent tensors; every gradient is complete here)
    // (loglinear whose dW stayed on the main stream -- small steps: the W, b update (6-8 us alone) stays there too; its fork
    //  and join cost the main queue 2 x 5.6 us for 25 us of side-stream work: round-5 timeline of the W3C settings)
    const bool side_small = !is_dp(m) && !m->timing.enabled && m->nstreams >= 2 && (is_vs(m) || m->ll_dw_side);
    hipStream_t ss = side_small ? m->stream2 : m->stream;
    if (side_small && m->lazy_join && fork_late_mode(m)) {
        // (everything the small tensors need was issued on the side stream itself)
    } else if (side_small && m->lazy_join) {
        SERT_HIP(hipStreamWaitEvent(ss, m->ev_dense, 0));
    } else if (side_small) {
        SERT_HIP(hipEventRecord(m->ev_opt_fork, m->stream));
        SERT_HIP(hipStreamWaitEvent(ss, m->ev_opt_fork, 0));
    }
    // ---- the big tensors, in the reference's parameter order (models.py:542-543, :1105; the
    // tensors are independent): one streaming launch each -- or, data parallel, one launch per
    // owned piece as soon as its gradient slab has been reduce-scattered, the all-gather of the
    // updated slab right behind it
    bool any_ag = false;
    const int tail_splits = m->tail_splits;
    m->tail_splits = 0;
    // side-heavy schedule: the entity table is updated BEHIND the join of the tail (see below)
    static const bool no_defer = knob("SERT_RE_DEFER") && atoi(knob("SERT_RE_DEFER")) == 0;
    const bool defer_re = !no_defer && m->side_heavy && side_small && tail_splits > 0 && m->pt_big[1] && is_vs(m) && !c.keep_grads;
    // ... and so is a SMALL entity table (C2: 1000 x 128, one optimizer_small launch behind the entity chain on the side
    // stream).  Round 4, from the GPU timeline: that launch -- 4 us alone, 12-31 us beside the word table's Adam -- ended
    // when the Adam did, and the tail started 13 us later, behind the cross-queue join.  The tail needs nothing of it but
    // the sums of squares of R_e, which the PREVIOUS step's launch leaves (of the values it writes: same shares, same
    // order, the same bits -- sumsq_new_partial); the update itself only has to land before the next loss kernel
    // (settle_entity_update), so the main stream no longer joins the side stream at the end of a step.
    const bool defer_small = !no_defer && !defer_re && side_small && tail_splits > 0 && !m->pt_big[1] && is_vs(m) && !is_fs(m) &&
                             !c.keep_grads && m->lazy_join && fork_late_mode(m) && m->n_re > 0;
    const int re_cur = (int)(m->step & 1), re_nxt = re_cur ^ 1;
    const size_t re_cap = (size_t)2 * kOptBlocks;
    bool small_needs_join = !defer_small;
    int re_sq_lo = 0, re_nb = 0;
    for (int i = 0; i < 4; ++i) {
        if (!m->pt_big[i]) continue;
        const ParamTensor t = param_tensor(m, i);
        if (t.n == 0) continue;
        const int tg = i == 0 ? TG_OPT_WORD : TG_OPTIMIZER;
        if (i == 1 && defer_re) {
            // (its slots in the partial array stay where they are: the tail reads them from re_sq)
            const int64_t max_nb = t.n >= ((size_t)1 << 24) ? 2 * kOptBlocks : kOptBlocks;
            re_nb = (int)std::min<int64_t>(max_nb, cdiv(cdiv(t.n, 4), 256));
            re_sq_lo = n_sq;
            n_sq += re_nb;
            continue;
        }
        if (!(is_dp(m) && m->pt_sharded[i])) {
            ScopedTimer tm(m, tg);
            // (tables of 2^24 elements and more -- C4: 150 M and 30 M -- stream faster over twice the
            //  workgroups: word table 708 -> 644 us, entity table 180 -> 167 us; no difference at
            //  C2's 12.8 M.  The count depends on the tensor size only: same tree in every run.)
            const int64_t max_nb = t.n >= ((size_t)1 << 24) ? 2 * kOptBlocks : kOptBlocks;
            const int nb = (int)std::min<int64_t>(max_nb, cdiv(cdiv(t.n, 4), 256));
            const uint32_t* tf = (i == 0 && m->use_touched) ? bits : nullptr;
            // Where it pays: the catch-up loop costs VALU time, the saving is the rows NOT written.  Measured (round 4,
            // tools/experiments/r04_lazy_sweep.sh; word-table update alone / step): C4, a batch touches 14 % of the rows:
            // 639 -> 530 us / 1.89 -> 1.77 ms; the reference's product-search settings (batch 4096, 12 %): 113 -> 85 us /
            // 231 -> 207 us; its W3C loglinear settings (batch 1024, 5 %): 141 -> 79 us / 331 -> 263 us; C2 dims at batch
            // 16384 (20 %): 58.5 -> 48 us; at C2's own batch (44 %, this or the next batch 69 %) a draw: 61.5 -> 57.5 us
            // alone, the step equal -- dense there.  Lazy up to a touched fraction of 0.35 (SERT_LAZY_MAX in a
            // variants build: 0 = never, 1 = always).
            // Round 5: dense_update_skip does not read the rows nobody needs, and then the lazy form wins at C2 too (a batch
            // touches 44 % of the rows, this or the next one 69 %: 61 -> 53 us per launch on average, 0.271 -> 0.261 ms per
            // step, tools/experiments/r05_skip_c2.sh).  With an announced next batch: lazy up to m->lazy_max (SERT_LAZY_MAX,
            // default 0.5 -- above that nearly every row is needed by this batch or the next and the passes that read
            // everything carry the predictions for nothing); without one every lazy step reads and writes every row, and
            // the dense launch takes over above 0.35 as before.
            const float lazy_max = (next_bits && m->lazy_skip) ? m->lazy_max : std::min(m->lazy_max, 0.35f);
            // (and for tables of 4 M elements and more: a small one lives in the caches, where the dense launch costs
            //  nothing to save -- the reference's C1, 640 k parameters: 91 us dense, 95 us lazy)
            static const bool lazy_small = variant_knob("SERT_LAZY_SMALL_TABLES") != nullptr;
            if (i == 0 && tf && !m->early_issued && m->rw_last[0] && !c.keep_grads && c.word_dim % 4 == 0 &&
                m->cur_touched_frac <= lazy_max && (t.n >= ((size_t)1 << 22) || lazy_small)) {
                // the LAZY form of the dense update (kernels_opt.h): rows neither this batch nor the announced next one
                // touches are read (their share of sum(p^2)) but not written, except every kLazyK-th update
                LazyArgs lz = lazy_args(m, m->step - 1, /*update=*/1);
                int nb_skip = 0;    // (dense_update_skip's own grid, when it takes the launch)
                lz.next_bits = next_bits;
                lz.write_all = (next_bits == nullptr || m->step % kLazyK == 0) ? 1 : 0;
                const unsigned d4 = (unsigned)c.word_dim / 4;
                if (m->lazy_skip && m->rw_pred && d4 <= 256) {
                    // rows nobody needs are not read either (kernels_opt.h: dense_update_skip): a full pass when there are no
                    // valid predictions (first lazy step, behind a dense step or new parameters, no hint) and kLazyK updates
                    // after the last one
                    const int64_t u = m->step;                      // the update being applied
                    const bool sparse = next_bits && m->rw_pred_ok && u < m->rw_pred_T;
                    lz.write_all = sparse ? 0 : 1;
                    if (!sparse) m->rw_pred_T = u + kLazyK;
                    SkipArgs sk;
                    sk.pred = m->rw_pred;
                    sk.stride = m->rw_pred_stride;
                    sk.npred = next_bits ? (int)(m->rw_pred_T - 1 - u) : 0;
                    for (int j = 0; j < kLazyK; ++j) {
                        AdamArgs a2; AdadeltaArgs d2;
                        optimizer_args(m, u + 1 + j, &a2, &d2);
                        sk.a_fut[j] = a2.a_t;
                    }
                    m->rw_pred_ok = next_bits != nullptr;
                    const unsigned nrows = (unsigned)c.vocab_size;
                    // (its own grid: a workgroup walks its rows one lane group per row, so shorter row ranges balance better --
                    //  SERT_SKIP_BLOCKS in a variants build, tools/experiments/r05_skip_blocks.sh)
                    static const int skip_blocks_knob = variant_knob("SERT_SKIP_BLOCKS") ? atoi(variant_knob("SERT_SKIP_BLOCKS")) : 0;
                    const int nb_dense = nb;
                    const int nb = std::max(1, std::min<int>(skip_blocks_knob > 0 ? std::min(skip_blocks_knob, 4 * kOptBlocks) : skip_grid(nb_dense, t.n),
                                                             (int)cdiv(nrows, 8u)));
                    nb_skip = nb;
#define SERT_SKIP_LAUNCH(ADAM, LPR, CPL)                                                                                   \
    hipLaunchKernelGGL((dense_update_skip<ADAM, LPR, CPL>), dim3(nb), dim3(256), 0, m->stream, t.p, (const float*)t.g, t.s0, \
                       t.s1, nrows, aa, da, m->red_sq + n_sq, tf, (unsigned)c.word_dim, lz, sk)
                    // d_w = 300: 75 float4 per row are 3 x 32 lanes at 78 % or 2 x 64 at 59 % of the lanes, and eight rows per
                    // workgroup instead of four.  Measured (tools/experiments/r05_skip_32x3.sh, three rounds, 64 x 2 -> 32 x 3):
                    // product-search settings 0.1756 -> 0.1704 ms (the launch alone 68.5 us either way), W3C loglinear settings
                    // 0.1922 -> 0.1894 (60.1 -> 57.5 us), C4's 150 M-element table 1.348 -> 1.344 (381 -> 387 us): taken below
                    // 2^26 elements.  SERT_SKIP_32X3=0 / 1 (variants build) forces it off / on.
                    static const int skip_32x3_knob = variant_knob("SERT_SKIP_32X3") ? atoi(variant_knob("SERT_SKIP_32X3")) : -1;
                    const bool skip_32x3 = skip_32x3_knob >= 0 ? skip_32x3_knob != 0 : t.n < ((size_t)1 << 26);
#define SERT_SKIP_SHAPE(ADAM)                                                                   \
    do {                                                                                        \
        if (d4 <= 32) { SERT_SKIP_LAUNCH(ADAM, 32, 1); ++m->upd_counts[2]; }                    \
        else if (d4 <= 64) { SERT_SKIP_LAUNCH(ADAM, 64, 1); ++m->upd_counts[3]; }               \
        else if (d4 <= 96 && skip_32x3) { SERT_SKIP_LAUNCH(ADAM, 32, 3); ++m->upd_counts[4]; }  \
        else if (d4 <= 128) { SERT_SKIP_LAUNCH(ADAM, 64, 2); ++m->upd_counts[5]; }              \
        else if (d4 <= 192) { SERT_SKIP_LAUNCH(ADAM, 64, 3); ++m->upd_counts[6]; }              \
        else { SERT_SKIP_LAUNCH(ADAM, 64, 4); ++m->upd_counts[7]; }                             \
        ++m->upd_counts[sparse ? 9 : 8];                                                        \
    } while (0)
                    if (is_vs(m)) SERT_SKIP_SHAPE(true);
                    else SERT_SKIP_SHAPE(false);
#undef SERT_SKIP_SHAPE
#undef SERT_SKIP_LAUNCH
                } else {
                ++m->upd_counts[1];
                if (is_vs(m))
                    hipLaunchKernelGGL((dense_update_lazy<true>), dim3(nb), dim3(256), 0, m->stream, t.p, (const float*)t.g, t.s0, t.s1, t.n,
                                       aa, da, m->red_sq + n_sq, tf, (unsigned)c.word_dim, lz);
                else
                    hipLaunchKernelGGL((dense_update_lazy<false>), dim3(nb), dim3(256), 0, m->stream, t.p, (const float*)t.g, t.s0, t.s1, t.n,
                                       aa, da, m->red_sq + n_sq, tf, (unsigned)c.word_dim, lz);
                }
                m->rw_last_cur ^= 1;
                m->rw_stale = !lz.write_all;
                m->rw_ready_batch = lz.write_all ? -1 : m->lazy_next;
                n_sq += nb_skip > 0 ? nb_skip : nb;
                continue;
            }
            if (i == 0) m->rw_pred_ok = false;   // (a dense launch moves every row off its predicted trajectory)
            if (i == 0) ++m->upd_counts[0];
            if (i == 0) SERT_TRY(ensure_rw_current(m, -1, m->step - 1));    // (a dense launch assumes every row is at the previous step; m->step is already this update's number)
            // (side_heavy: the entity table streams on the side stream, behind its gradient chain.  Loglinear with dW on the
            //  side stream: a W large enough to be a "big tensor" -- d x V_e >= 2^22, C4 -- is updated THERE, behind dW and
            //  its combine; on the main stream its update read dW's gradient while the side stream was still writing it:
            //  two runs of the C4 loglinear step differed by 0.5 % in W after two steps, tools/experiments/r04_ll_c4_rep.py)
            const bool on_side = side_small && ((i == 1 && m->side_heavy) || (i >= 2 && m->ll_dw_side));
            launch_stream_opt(m, on_side ? ss : m->stream, t.p, t.g, t.s0, t.s1, t.n, nb, aa, da,
                              m->red_sq + n_sq, tf,
                              i == 0 ? (unsigned)c.word_dim : 1u,
                              (i == 0 && tf && m->early_issued) ? kRowsTouched : kRowsAll);
            n_sq += nb;
            continue;
        }
        const size_t sc = m->pt_sc[i];
        const int nch = m->ar_chunks;
        // word table owned by rows: the gradient rows were returned to their owners (xr_return_grads);
        // rows of the owned range no rank touched take a zero gradient without reading it, and nothing
        // is gathered -- the next forward fetches the rows it needs (xr_fetch_params)
        const bool by_rows = (i == 0) && m->xr_on;
        {
            ScopedTimer tm(m, tg);
            for (int ch = 0; ch < nch; ++ch) {
                if (exchanged && m->rs_issued[i]) SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_rs_done[i][by_rows ? 0 : ch], 0));
                const size_t off = piece_off(m, i, ch);
                const int nb = (int)std::min<int64_t>(std::max(1, kOptBlocks / nch), cdiv(cdiv(sc, 4), 256));
                // (the padding behind the tensor's last element is zero with a zero gradient: it stays zero)
                const uint32_t* ub = by_rows ? m->xr_ubits + (size_t)m->xr_batch * (size_t)m->xr->owned_bit_words : nullptr;
                launch_stream_opt(m, m->stream, t.p + off, t.g + off, t.s0 + (size_t)ch * sc, t.s1 + (size_t)ch * sc, sc,
                                  nb, aa, da, m->sq_scratch, ub, by_rows ? (unsigned)c.word_dim : 1u);
                if (by_rows) {
                    // (recorded in every mode: a later step on the communication stream waits for it whatever
                    //  mode THIS step ran in)
                    if (m->ev_word_updated) SERT_HIP(hipEventRecord(m->ev_word_updated, m->stream));
                    m->rw_full = false;
                    m->xr_fetched_batch = -1;
                    continue;
                }
                m->comm_bytes_moved += 8.0 * (double)sc * (double)(m->world - 1);   // all-gather: (N-1) pieces out, (N-1) in
                if (exchanged) {
                    SERT_HIP(hipEventRecord(m->ev_opt_done[i][ch], m->stream));
                    SERT_HIP(hipStreamWaitEvent(m->comm_stream, m->ev_opt_done[i][ch], 0));
                    SERT_NCCL(g_rccl.AllGather(t.p + off, t.p + (size_t)ch * slab_elems(m, i), sc, /*ncclFloat32*/ 7,
                                               m->comm, m->comm_stream));
                    any_ag = true;
                }
            }
        }
        m->rs_issued[i] = false;
        if (by_rows) {
            // (nothing to gather)
        } else if (m->host_ar) {
            ScopedTimer tm(m, TG_ALLGATHER);
            for (int ch = 0; ch < nch; ++ch) SERT_TRY(host_allgather(m, t.p + (size_t)ch * slab_elems(m, i), sc, m->stream));
        } else if (m->comm && m->timing.enabled) {
            ScopedTimer tm(m, TG_ALLGATHER);
            for (int ch = 0; ch < nch; ++ch)
                SERT_NCCL(g_rccl.AllGather(t.p + piece_off(m, i, ch), t.p + (size_t)ch * slab_elems(m, i), sc, 7, m->comm,
                                           m->stream));
        }
    }
    if (exchanged) SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_ar_done, 0));
    if (side_small && egrad_writes_every_row(m) && c.kind == SERT_KIND_VECTORSPACE && !is_fs(m) && c.num_negatives > 0 && !c.keep_grads &&
        m->neg_alt_step != m->step) {
        // the next step's negatives (Philox position = the step counter after this update), drawn
        // here on the side stream unless the step's forward already drew them in front of its fork:
        // ev_small below orders them before anything of the next step
        const int64_t count = (int64_t)c.batch_size * c.num_negatives;
        hipLaunchKernelGGL(vs_sample_negatives, dim3(grid_for((count + 3) / 4)), dim3(256), 0, ss, m->neg_alt, count,
                           (int64_t)m->rank * count, (uint32_t)c.num_entities, c.seed, (uint64_t)m->step * 2);
        m->neg_alt_step = m->step;
    }
    // Late fork (fork_late_mode): dW / db were produced on the main stream, dR_e on the side
    // stream -- W and b are updated on the main stream, R_e on the side stream, no event between.
    const bool split_small = side_small && m->lazy_join && fork_late_mode(m);
    auto small_tensors = [&](hipStream_t ss, unsigned mask) {
        // everything small goes into one launch (a kernel boundary costs more than updating it)
        ScopedTimer t(m, TG_OPTIMIZER);
        SmallTensors st;
        int k = 0, blocks = 0;
        for (int i = 1; i < 4; ++i) {
            const ParamTensor t2 = param_tensor(m, i);
            if (m->pt_big[i] || t2.n == 0 || !((mask >> i) & 1u)) continue;
            st.p[k] = t2.p; st.g[k] = t2.g; st.s0[k] = t2.s0; st.s1[k] = t2.s1; st.count[k] = t2.n;
            const bool parts = (i == 1) && m->re_in_parts;
            st.gparts[k] = parts ? m->epart : nullptr;
            st.ngroups[k] = parts ? m->eg_groups : 0;
            st.gstride[k] = parts ? (unsigned long long)m->n_re : 0ull;
            st.l2k[k] = t2.l2 ? l2k : 0.f;
            st.first_block[k] = blocks;
            blocks += (int)std::min<int64_t>(512, cdiv(t2.n, 256));
            ++k;
        }
        for (int i = k; i < 3; ++i) { st.p[i] = st.g[i] = st.s0[i] = st.s1[i] = nullptr; st.count[i] = 0; st.l2k[i] = 0.f; st.gparts[i] = nullptr; st.ngroups[i] = 0; st.gstride[i] = 0; }
        for (int i = k; i <= 3; ++i) st.first_block[i] = blocks;
        float* sq = m->red_sq + n_sq;
        float* sq_new = nullptr;
        if (defer_small && mask == 0x2u && blocks > 0) {
            // (R_e alone in this launch: its partial slots [n_sq, n_sq + blocks) are read from re_sq by the tail)
            re_sq_lo = n_sq;
            re_nb = blocks;
            sq_new = m->re_sq + re_nxt * re_cap;
            if (m->re_sq_for[re_cur] != m->step) {
                // no previous launch left this step's sums (first step, another schedule in between, the host replaced
                // the table): the same partials from a read-only pass IN FRONT of the update, and the tail joins once
                hipLaunchKernelGGL(sumsq_like_small, dim3(blocks), dim3(256), 0, ss, (const float*)m->re, m->n_re, l2k,
                                   m->re_sq + re_cur * re_cap);
                m->re_sq_for[re_cur] = m->step;
                small_needs_join = true;
            }
        }
        if (blocks > 0) {
            const bool keep = c.keep_grads != 0;
            if (is_vs(m) && sq_new) {
                hipLaunchKernelGGL((optimizer_small<true, false>), dim3(blocks), dim3(256), 0, ss, st, aa, da, sq, sq_new);
            } else if (is_vs(m)) {
                if (keep) hipLaunchKernelGGL((optimizer_small<true, true>), dim3(blocks), dim3(256), 0, ss, st, aa, da, sq);
                else      hipLaunchKernelGGL((optimizer_small<true, false>), dim3(blocks), dim3(256), 0, ss, st, aa, da, sq);
            } else {
                if (keep) hipLaunchKernelGGL((optimizer_small<false, true>), dim3(blocks), dim3(256), 0, ss, st, aa, da, sq);
                else      hipLaunchKernelGGL((optimizer_small<false, false>), dim3(blocks), dim3(256), 0, ss, st, aa, da, sq);
            }
        }
        n_sq += blocks;
    };
    if (tail_splits > 0) {
        small_tensors(ss, 0x2u);          // R_e; W and b are updated by the tail launch below
    } else if (split_small && dw_third_queue(m)) {
        small_tensors(m->stream3, 0xCu);  // W, b: behind dW and its combine on the third queue
        SERT_HIP(hipEventRecord(m->ev_join3, m->stream3));
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_join3, 0));
        small_tensors(ss, 0x2u);          // R_e
    } else if (split_small) {
        small_tensors(m->stream, 0xCu);   // W, b
        small_tensors(ss, 0x2u);          // R_e
    } else {
        small_tensors(ss, 0xEu);
    }
    if (defer_re && m->re_sq_for[re_cur] != m->step) {
        // no previous deferred launch left this step's sums (first step, another schedule in between, the
        // host replaced the table): the same partials from a read-only pass, in front of the join
        hipLaunchKernelGGL(sumsq_like_adam, dim3(re_nb), dim3(256), 0, ss, (const float*)m->re, m->n_re, m->re_sq + re_cur * re_cap);
        m->re_sq_for[re_cur] = m->step;
    }
    if (side_small && small_needs_join) {
        SERT_HIP(hipEventRecord(m->ev_small, ss));
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_small, 0));
    }
    if (defer_small && re_nb > 0) {
        // the update is in the side stream; the next reader of R_e (and, through this stream's order, of the negatives
        // drawn in front of this step's fork) waits for it in settle_entity_update
        m->re_sq_for[re_nxt] = m->step + 1;
        SERT_HIP(hipEventRecord(m->ev_re, ss));
        m->re_pending = true;
    }
    if (defer_re) {
        // The entity table's L2 + Adam, behind the join: the tail does not wait for it.  Nothing reads R_e,
        // its state or dR_e before the next loss kernel (settle_entity_update), so its 0.96 GB stream beside
        // the tail and the next step's gather and projection GEMM.  It also leaves the sums of squares of the
        // UPDATED table: the next step's regularisation term.
        const ParamTensor t = param_tensor(m, 1);
        launch_stream_opt(m, ss, t.p, t.g, t.s0, t.s1, t.n, re_nb, aa, da, m->red_sq + re_sq_lo, nullptr, 1u, kRowsAll,
                          m->re_sq + re_nxt * re_cap);
        m->re_sq_for[re_nxt] = m->step + 1;
        SERT_HIP(hipEventRecord(m->ev_re, ss));
        m->re_pending = true;
    }
    if (m->early_issued) {   // the untouched rows' sum of squares (and their update) must have landed
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_early, 0));
        m->early_issued = false;
    }
    {
        ScopedTimer t(m, TG_FINALIZE);
        const float inv_batch = 1.0f / (float)c.global_batch_size;
        const float reg_scale = c.lambda_ > 0.f ? c.lambda_ / (2.0f * (float)c.global_batch_size) : 0.f;
        // single GPU: the loss partials directly; data parallel: the all-reduced scalars (loss
        // sum; sum of squares of the sharded tensors -- the replicated ones come from red_sq)
        const float* lp = is_dp(m) ? m->g_loss : (m->loss_from_rows ? m->rowloss : m->red_loss);
        const int nl = is_dp(m) ? 1 : n_loss_partials;
        unsigned* flag = publish ? reinterpret_cast<unsigned*>(loss_dst + 4) : nullptr;
        // timing knock-out (variants build, WRONG loss: the sums of squares of the word table are read while its update runs):
        // the tail on the SIDE stream behind the entity chain -- what taking it off the main queue would buy (r06 experiments, item 6)
        static const bool ko_tail_side = variant_knob("SERT_KO_TAIL_SIDE") != nullptr;
        const bool tail_side = ko_tail_side && tail_splits > 0 && m->dw_side_first && defer_small && side_small;
        hipStream_t ts = tail_side ? ss : m->stream;
        if (tail_splits > 0 && m->tail_early) {
            // (knock-out: W and b were updated behind dW on the side stream; only the loss is left -- the next projection waits for ev_dense)
            hipLaunchKernelGGL(finalize_loss, dim3(1), dim3(256), 0, m->stream, lp, nl, m->red_sq, n_sq, inv_batch, reg_scale, loss_dst, flag,
                               publish ? ++m->loss_seq : 0u, (const float*)nullptr);
            m->w_early_pending = true;
        } else
        if (tail_splits > 0) {
            if (m->dw_side_first && !tail_side) SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_dense, 0));   // (dW / db slabs: side stream)
            TailArgs ta;
            ta.part = m->tail_part ? m->tail_part : m->part; ta.splits = tail_splits; ta.stride = m->tail_stride;
            ta.W = m->W; ta.b = m->b; ta.s0_w = m->s0_w; ta.s1_w = m->s1_w; ta.s0_b = m->s0_b; ta.s1_b = m->s1_b;
            ta.g_w = m->g_w; ta.g_b = m->g_b;
            ta.n_w = (unsigned)m->n_w; ta.n_b = (unsigned)m->n_b;
            ta.aa = aa;
            ta.loss_partials = lp; ta.n_loss = nl;
            ta.sq_partials = m->red_sq; ta.n_sq = n_sq;
            const bool alt = defer_re || (defer_small && re_nb > 0);
            ta.sq_alt = alt ? m->re_sq + re_cur * re_cap : nullptr;
            ta.sq_alt_lo = alt ? re_sq_lo : 0;
            ta.sq_alt_hi = alt ? re_sq_lo + re_nb : 0;
            ta.inv_batch = inv_batch; ta.reg_scale = reg_scale;
            ta.out = loss_dst; ta.host_flag = flag; ta.seq = publish ? ++m->loss_seq : 0u;
            ta.blk = m->tail_blk;
            if (++m->tail_launch_seq == 0) ++m->tail_launch_seq;   // (0 = "never written")
            ta.launch_seq = m->tail_launch_seq;
            const int nb = cdiv((int64_t)(m->n_w + m->n_b), 64);
            if (c.keep_grads) hipLaunchKernelGGL((vs_tail<true>), dim3(nb), dim3(1024), 0, ts, ta);
            else              hipLaunchKernelGGL((vs_tail<false>), dim3(nb), dim3(1024), 0, ts, ta);
            if (tail_side) {     // (the next projection reads W: it waits for this, see step_forward_backward)
                SERT_HIP(hipEventRecord(m->ev_re, ss));
                m->re_pending = true;
                m->w_pending = true;
            }
        } else
        hipLaunchKernelGGL(finalize_loss, dim3(1), dim3(256), 0, m->stream, lp, nl, m->red_sq,
                           n_sq, inv_batch, reg_scale, loss_dst, flag, publish ? ++m->loss_seq : 0u,
                           is_dp(m) ? (const float*)m->g_sq : (const float*)nullptr);
    }
    // (opt-in split optimiser: the next step's untouched-row launch, on its own stream, may start
    // once this step has read its sum-of-squares partials and updated the rows it owns)
    if (adam_split_enabled() && !is_dp(m) && !m->timing.enabled) SERT_HIP(hipEventRecord(m->ev_word_opt, m->stream));
    if (any_ag) {
        // the loss leaves first; the next kernel that reads a parameter waits for the last slab
        SERT_HIP(hipEventRecord(m->ev_ag_done, m->comm_stream));
        SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_ag_done, 0));
    }
    return 0;
}

// First element of the gradient buffer the step's prologue has to zero when the word table needs no
// zeroing: behind g_rw -- and behind g_re too where the sorted entity-gradient chain runs, which writes
// EVERY row of dR_e (a run inside one chunk by the chunk, a longer one by the fix-up, an entity without a
// pair as zeros by the fix-up): 120 MB less to write per step at C4.
static size_t zero_from(const sert_model* m) {
    static const bool all = variant_knob("SERT_ZERO_GRE") != nullptr;   // cross-check knob
    if (!all && m->cfg.kind == SERT_KIND_VECTORSPACE && !m->epart && m->n_re > 0 && m->g_re == m->gflat + m->ar_split)
        return m->ar_split + round_up(m->pt_pad[1], 4);
    return m->ar_split;
}

// The entity-gradient chain writes EVERY row of dR_e (the sort-free path by construction, the sorted one through its fix-up):
// a step whose negatives were drawn ahead then needs no prologue launch at all.
static bool egrad_writes_every_row(const sert_model* m) { return m->epart != nullptr || zero_from(m) > m->ar_split; }

static bool fused_prologue_applies_with(const sert_model* m, bool touched) {
    return is_vs(m) && !is_fs(m) && !m->timing.enabled && m->nstreams >= 2 && touched &&
           m->cfg.num_negatives > 0 && (m->gflat_alloc - zero_from(m)) % 4 == 0;
}
static bool fused_prologue_applies(const sert_model* m) { return fused_prologue_applies_with(m, m->use_touched); }
// sampler of optimiser step m->step + zeroing of the small gradient buffers
static void launch_fused_prologue(sert_model* m) {
    const int64_t count = (int64_t)m->cfg.batch_size * m->cfg.num_negatives;
    hipLaunchKernelGGL(vs_sample_negatives, dim3(grid_for((count + 3) / 4)), dim3(256), 0, m->stream, m->neg,
                       count, (int64_t)m->rank * count, (uint32_t)m->cfg.num_entities, m->cfg.seed,
                       (uint64_t)m->step * 2, reinterpret_cast<float4*>(m->gflat + zero_from(m)),
                       (m->gflat_alloc - zero_from(m)) / 4, (uint4*)nullptr, (size_t)0);
}

static bool use_touched_now(const sert_model* m) {
    static const bool no_touched = knob("SERT_NO_TOUCHED") != nullptr;   // cross-check knob
    return !no_touched && !is_dp(m) && !m->cfg.keep_grads && m->cfg.word_dim % 4 == 0 &&
           m->n_rw < ((size_t)1 << 32) && m->split[SERT_SPLIT_TRAIN].idx_touched_bits != nullptr;
}

// Everything of a training step that does NOT change the model: forward, loss, backward into
// the gradient scratch (all of it a function of parameters, data and step counter only).
// `fused_pre_out`: whether the step took the fused main-stream prologue.
static int step_forward_backward(sert_model* m, const DataSplit& ds, int64_t batch_index,
                                 const int64_t* negatives, bool* fused_pre_out) {
    // Single GPU: the word-gradient table is not zeroed -- the segmented reduction flags
    // the rows it writes and the optimiser takes every other row's gradient as zero.
    // (Data parallel: the all-reduce needs the dense table; keep_grads: so does the caller.)
    m->use_touched = use_touched_now(m);
    SERT_TRY(ensure_rw_current(m, batch_index));   // (lazy word-table update: the rows this batch reads must be current)
    // Prologue (zeroing, negative sampling): nothing before the loss kernel needs it, so
    // for the vectorspace step it runs on the side stream beside gather + projection.
    m->lazy_join = false;
    m->ll_dw_side = false;
    m->dw_side_first = false;
    m->dp_late_join = false;
    m->side_heavy = false;
    const bool side_pre = is_vs(m) && !is_fs(m) && !m->timing.enabled && m->nstreams >= 2;
    // One fused prologue launch on the MAIN stream (sampler + zeroing of the small gradient
    // buffers and the row flags) when nothing big has to be zeroed and the device draws the
    // negatives: no side-stream prologue, no cross-queue wait in front of the loss kernel.
    // negatives drawn at the end of the previous step for exactly this step: nothing to do (the
    // sort-free entity-gradient path needs no zeroed buffer either: every row it owns is written)
    // (round 5: also behind the SORTED chain of a big entity table -- the product-search settings, C4 --, whose steps used to
    //  start with a sampler + zeroing launch on the main stream, 6 us + a queue bubble: nothing it zeroed is read there)
    const bool have_neg = negatives == nullptr && egrad_writes_every_row(m) && m->neg_alt_step == m->step && side_pre &&
                          fused_prologue_applies(m);
    if (have_neg) {
        std::swap(m->neg, m->neg_alt);
        m->neg_alt_step = -1;
    }
    m->neg_side_ready = have_neg;      // (this step's negatives are complete in the side stream's order: vs_backward)
    const bool fused_pre = side_pre && negatives == nullptr && fused_prologue_applies(m);
    hipStream_t pre = (side_pre && !fused_pre) ? m->stream2 : m->stream;
    *fused_pre_out = (pre == m->stream);   // no side-stream prologue: no end-of-step event needed
    if (pre != m->stream && m->step_done_pending) {
        // a side-stream prologue must follow the previous step, which (fused prologue) did not
        // mark its end: nothing of this step is on the main stream yet, so mark it now
        SERT_HIP(hipEventRecord(m->ev_step_done, m->stream));
        m->step_done_pending = false;
    }
    // data parallel, word table owned by rows: the rows this batch touches arrive from their owners
    if (m->xr_on) {
        m->xr_batch = batch_index;
        SERT_TRY(xr_fetch_params(m, batch_index));
    }
    // the main stream's first kernels go out BEFORE the prologue's host calls: the GPU
    // starts on gather + projection while the host is still enqueueing
    if (is_vs(m) && !is_fs(m)) {
        if (m->w_pending) { SERT_TRY(settle_entity_update(m)); m->w_pending = false; }   // (W, b updated on the side stream)
        if (m->w_early_pending) { SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_dense, 0)); m->w_early_pending = false; }
        if (m->projected_batch != batch_index) SERT_TRY(vs_project(m, ds, batch_index));
        m->projected_batch = -1;
    }
    SERT_TRY(settle_entity_update(m));   // (the prologue zeroes dR_e, the loss kernel reads R_e)
    if (have_neg) {
        // (no prologue launch at all)
    } else if (fused_pre) {
        launch_fused_prologue(m);
    } else {
        // (the previous step's optimiser and loss kernels read what the prologue overwrites)
        if (pre != m->stream) SERT_HIP(hipStreamWaitEvent(pre, m->ev_step_done, 0));
        if (m->use_touched && !is_vs(m) && !is_dp(m)) {
            // loglinear, single GPU: NOTHING to zero -- the word table's gradient is read through the touched-row bitmap, and
            // g_W / g_b are written in full by the split-K combine behind dW (ll_backward); the 5 us memset was a launch of
            // its own on the chain of an 18-kernel step (the reference's W3C settings: round-5 timeline)
        } else if (m->use_touched || m->xr_on) {
            // (by rows: dR_w is written where this rank's batch touches, read where the lists say, and
            //  the owned rows nobody touched are never read -- the table needs no zeroing)
            SERT_HIP(hipMemsetAsync(m->gflat + zero_from(m), 0, (m->gflat_alloc - zero_from(m)) * sizeof(float), pre));
        } else {
            SERT_HIP(hipMemsetAsync(m->gflat, 0, m->gflat_alloc * sizeof(float), pre));
        }
        // (vectorspace with a side-stream prologue: BEHIND the negatives and their event, see below)
        if (!(is_vs(m) && !is_fs(m) && pre != m->stream)) SERT_TRY(owned_sum_of_squares(m, pre));
    }
    if (is_fs(m)) {
        SERT_TRY(fs_forward<true>(m, ds, batch_index));
        SERT_TRY(fs_backward(m, ds, batch_index));
        SERT_TRY(reduce_rowloss(m, m->stream));
    } else if (is_vs(m)) {
        if (!fused_pre) SERT_TRY(vs_negatives(m, negatives, (uint64_t)m->step * 2, pre));
        if (pre != m->stream) {
            SERT_HIP(hipEventRecord(m->ev_neg, pre));
            SERT_HIP(hipStreamWaitEvent(m->stream, m->ev_neg, 0));
            // Data parallel: the sums of squares of the owned pieces (the regularisation term's share of this rank: a pass
            // over the owned slice of every sharded tensor + one combine) only feed the all-reduce of the replicated rest,
            // which joins this stream much later -- they used to sit IN FRONT of the negatives, and the loss kernel waited
            // for all of it (round-5 timeline at C2, world of one: the loss kernel 25 us behind the projection).
            if (!have_neg && !fused_pre) SERT_TRY(owned_sum_of_squares(m, pre));
        }
        SERT_TRY(vs_loss<true>(m, ds, batch_index));
        SERT_TRY(vs_backward(m, ds, batch_index));   // (loss partials: beside dW)
    } else {
        SERT_TRY(ll_forward<true>(m, ds, batch_index));
        SERT_TRY(ll_backward(m, ds, batch_index));
        SERT_TRY(reduce_rowloss(m, m->stream));
    }
    return 0;
}

// May the forward + backward of a hinted next batch run ahead of the host (before the loss
// of the current step has been read)?  It touches nothing but activations and gradient scratch.
// Single GPU: product settings, device sampler.  Data parallel (round 6): as well -- the run-ahead then includes the
// collectives of the backward (the gradient rows' all-to-all / the reduce-scatter), which is safe because every rank
// sees the same sequence of hints and batches (the global batch order is rank-invariant) and therefore
// issues, keeps or discards the same run-ahead; the parameter fetch of the hinted batch was already issued this way.
// Without it the data-parallel step was HOST-bound: its ~45 runtime calls only started when the previous loss had
// arrived (world of one, 8192 rows: 0.174 ms against 0.096 on one GPU; profiles/r06_experiments.txt item 4).
// SERT_DP_RUN_AHEAD=0 (variants build) restores the round-5 behaviour.
static bool can_speculate_step(const sert_model* m) {
    if (m->timing.enabled || m->cfg.keep_grads) return false;
    if (is_dp(m)) {
        static const bool dp_off = variant_knob("SERT_DP_RUN_AHEAD") && atoi(variant_knob("SERT_DP_RUN_AHEAD")) == 0;
        return !dp_off && is_vs(m) && !is_fs(m);
    }
    if (is_vs(m) && !is_fs(m)) return use_touched_now(m) && fused_prologue_applies_with(m, true);
    return true;   // loglinear / full-softmax: the whole step lives on the main stream
}

static int train_step_async(sert_model* m, int64_t batch_index, const int64_t* negatives,
                            float* loss_dst, bool publish = false) {
    const DataSplit& ds = m->split[SERT_SPLIT_TRAIN];
    const int B = m->cfg.batch_size;
    if (m->cfg.inference_only) SERT_FAIL("model was created inference_only");
    if (m->comm_dead) SERT_FAIL("the communicator of this data-parallel model was destroyed");
    if (ds.N == 0) SERT_FAIL("no training data uploaded");
    if (batch_index < 0 || (batch_index + 1) * (int64_t)B > ds.N) SERT_FAIL("batch_index out of range");
    if (m->instep.on) ++m->instep.steps;
    // what the previous call already ran ahead for this step (sert_hint_next_batch)
    const bool have_fb = negatives == nullptr && m->spec_fb_batch == batch_index && m->spec_fb_step == m->step &&
                         can_speculate_step(m);
    bool fused_pre = true;   // (a speculated step always ran its prologue on the main stream)
    if (have_fb) m->spec_fb_batch = -1;     // consumed
    else discard_run_ahead(m);              // a run-ahead for something else: discard it cleanly
    const uint32_t* bits = ds.idx_touched_bits ? ds.idx_touched_bits + (size_t)batch_index * ds.bit_words : nullptr;
    // this step is committed: the rows its batch does not touch are updated beside it
    if (!have_fb) m->use_touched = use_touched_now(m);
    SERT_TRY(issue_untouched_rows_update(m, bits));
    if (!have_fb) SERT_TRY(step_forward_backward(m, ds, batch_index, negatives, &fused_pre));
    SERT_TRY(allreduce_rest(m));
    m->cur_touched_frac = (size_t)batch_index < ds.idx_batches.size()
                              ? (float)ds.idx_batches[(size_t)batch_index].num_distinct / (float)m->cfg.vocab_size : 1.f;
    // the rows the ANNOUNCED next batch touches: the lazy word-table update keeps them current (sert_hint_next_batch)
    const uint32_t* next_bits = nullptr;
    if (ds.idx_touched_bits && m->lazy_next >= 0 && (m->lazy_next + 1) * (int64_t)B <= ds.N)
        next_bits = ds.idx_touched_bits + (size_t)m->lazy_next * ds.bit_words;
    SERT_TRY(optimizer_and_loss(m, loss_dst, publish, bits, next_bits));
    if (is_dp(m)) m->comm_steps += 1;
    SERT_HIP(hipGetLastError());   // a rejected launch (bad configuration) surfaces here, not as a hang
    // (an event record stalls its queue for ~6 us: steps with the fused prologue skip it)
    if (fused_pre) m->step_done_pending = true;
    else { SERT_HIP(hipEventRecord(m->ev_step_done, m->stream)); m->step_done_pending = false; }
    return 0;
}

}  // namespace sert

using namespace sert;

// =============================== C ABI ===========================================
extern "C" {

const char* sert_last_error(void) { return g_last_error.c_str(); }

int sert_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int sert_device_info(int device, char* buf, size_t buflen) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) {
        g_last_error = "hipGetDeviceProperties failed";
        return -1;
    }
    return snprintf(buf, buflen, "%s %s: %d CUs, %.0f MHz, %.1f GiB, LDS/block %zu KiB", p.name,
                    p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000.0,
                    p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), p.sharedMemPerBlock / 1024);
}

static int create_resources(sert_model* m);
static int layout_gradients(sert_model* m);

// Data parallel: give every big tensor its ZeRO-1 geometry for (world, ar_chunks): parameters and
// gradients are re-allocated with the padding that makes the slabs equal, the optimiser state
// shrinks to the owned pieces (whatever it held so far is kept).
static int shard_setup(sert_model* m) {
    if (m->cfg.inference_only) SERT_FAIL("model was created inference_only");
    for (int i = 0; i < 4; ++i)
        if (m->pt_sharded[i]) SERT_FAIL("this model already has a data-parallel communicator");
    hipStream_t s = m->stream;
    SERT_TRY(ensure_rw_current(m, -1));    // (data parallel runs the dense update: no row may be behind)
    SERT_HIP(hipStreamSynchronize(s));
    SERT_HIP(hipStreamSynchronize(m->stream2));
    if (m->stream3) SERT_HIP(hipStreamSynchronize(m->stream3));
    float** P[4] = {&m->rw, &m->re, &m->W, &m->b};
    float** S0[4] = {&m->s0_rw, &m->s0_re, &m->s0_w, &m->s0_b};
    float** S1[4] = {&m->s1_rw, &m->s1_re, &m->s1_w, &m->s1_b};
    const size_t n[4] = {m->n_rw, m->n_re, m->n_w, m->n_b};
    for (int i = 0; i < 4; ++i) {
        if (!m->pt_big[i]) continue;
        const size_t unit = (size_t)m->world * (size_t)m->ar_chunks;
        size_t sc = round_up((n[i] + unit - 1) / unit, 64);
        if (i == 0 && m->xr_mode) {
            // owned by rows: a piece is a whole number of rows (a multiple of 16, so that it stays a
            // multiple of 64 elements)
            m->xr_rows_per_rank = (int64_t)round_up(cdiv((int64_t)m->cfg.vocab_size, m->world), 16);
            sc = (size_t)m->xr_rows_per_rank * (size_t)m->cfg.word_dim;
        }
        const size_t pad = sc * unit;
        float *np = nullptr, *ns0 = nullptr, *ns1 = nullptr;
        SERT_TRY(dzalloc(&np, pad, s));
        SERT_HIP(hipMemcpyAsync(np, *P[i], n[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
        SERT_TRY(dzalloc(&ns0, sc * m->ar_chunks, s));
        SERT_TRY(dzalloc(&ns1, sc * m->ar_chunks, s));
        for (int c = 0; c < m->ar_chunks; ++c) {
            const size_t off = (size_t)c * sc * m->world + (size_t)m->rank * sc;
            if (off >= n[i]) continue;
            const size_t cnt = std::min(sc, n[i] - off);
            SERT_HIP(hipMemcpyAsync(ns0 + (size_t)c * sc, *S0[i] + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
            SERT_HIP(hipMemcpyAsync(ns1 + (size_t)c * sc, *S1[i] + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        SERT_HIP(hipStreamSynchronize(s));
        (void)hipFree(*P[i]); (void)hipFree(*S0[i]); (void)hipFree(*S1[i]);
        *P[i] = np; *S0[i] = ns0; *S1[i] = ns1;
        m->pt_pad[i] = pad;
        m->pt_sc[i] = sc;
        m->pt_sharded[i] = true;
    }
    return layout_gradients(m);
}

// Optimiser state of a sharded tensor <-> a full-size host array.  get is COLLECTIVE (every rank
// calls it, in the same order): the owned pieces are all-gathered into a scratch tensor.
static int sharded_state_io(sert_model* m, int i, int k, float* host_out, const float* host_in) {
    const ParamTensor t = param_tensor(m, i);
    float* st = k == 0 ? t.s0 : t.s1;
    const size_t sc = m->pt_sc[i], pad = m->pt_pad[i], n = t.n;
    hipStream_t s = m->stream;
    if (host_in) {
        for (int c = 0; c < m->ar_chunks; ++c) {
            const size_t off = piece_off(m, i, c);
            SERT_HIP(hipMemsetAsync(st + (size_t)c * sc, 0, sc * sizeof(float), s));
            if (off < n)
                SERT_HIP(hipMemcpyAsync(st + (size_t)c * sc, host_in + off, std::min(sc, n - off) * sizeof(float),
                                        hipMemcpyHostToDevice, s));
        }
        SERT_HIP(hipStreamSynchronize(s));
        return 0;
    }
    if (m->comm_dead) SERT_FAIL("the communicator of this data-parallel model was destroyed");
    float* full = nullptr;
    SERT_TRY(dzalloc(&full, pad, s));
    for (int c = 0; c < m->ar_chunks; ++c)
        SERT_HIP(hipMemcpyAsync(full + piece_off(m, i, c), st + (size_t)c * sc, sc * sizeof(float), hipMemcpyDeviceToDevice, s));
    int rc = 0;
    if (m->host_ar) {
        for (int c = 0; c < m->ar_chunks && rc == 0; ++c) rc = host_allgather(m, full + (size_t)c * slab_elems(m, i), sc, s);
    } else {
        for (int c = 0; c < m->ar_chunks && rc == 0; ++c)
            if (g_rccl.AllGather(full + piece_off(m, i, c), full + (size_t)c * slab_elems(m, i), sc, 7, m->comm, s) != 0)
                rc = fail(__FILE__, __LINE__, "ncclAllGather failed (optimiser state)");
    }
    if (rc == 0 && hipMemcpyAsync(host_out, full, n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess)
        rc = fail(__FILE__, __LINE__, "copy of the gathered optimiser state failed");
    (void)hipStreamSynchronize(s);
    (void)hipFree(full);
    return rc;
}

// The events that order this model's own streams against each other.  device_scope: without the SYSTEM-scope release
// (hipEventDisableSystemFence) -- right while every consumer is a kernel on this device.  A model with a communicator
// (sert_comm_init / sert_comm_init_host) re-creates them with HIP's default flags: there ev_join / ev_dense / ev_fork also
// stand in front of a D2H copy (host transport) or a collective whose readers are peers, and an agent-scope release does
// not make the gradients visible to those (advisor, round 5).
static int create_intra_events(sert_model* m, bool device_scope) {
    hipEvent_t* evs[] = {&m->ev_word_opt, &m->ev_early, &m->ev_join3, &m->ev_step_done, &m->ev_neg, &m->ev_opt_fork,
                         &m->ev_dense, &m->ev_small, &m->ev_re, &m->ev_fork, &m->ev_join};
    const unsigned flags = hipEventDisableTiming | (device_scope ? (unsigned)hipEventDisableSystemFence : 0u);
    for (hipEvent_t* e : evs) {
        if (*e) { SERT_HIP(hipEventDestroy(*e)); *e = nullptr; }
        SERT_HIP(hipEventCreateWithFlags(e, flags));
    }
    m->events_device_scope = device_scope;
    return 0;
}

int sert_create(const sert_config* cfg, sert_model** out) {
    if (!cfg || !out) SERT_FAIL("null argument");
    if (cfg->struct_size != sizeof(sert_config)) SERT_FAIL("sert_config size mismatch (ABI)");
    if (cfg->kind != SERT_KIND_LOGLINEAR && cfg->kind != SERT_KIND_VECTORSPACE &&
        cfg->kind != SERT_KIND_VECTORSPACE_SOFTMAX)
        SERT_FAIL("bad kind");
    if (cfg->batch_size <= 0 || cfg->window_size <= 0 || cfg->vocab_size <= 0 ||
        cfg->num_entities <= 0 || cfg->word_dim <= 0)
        SERT_FAIL("sizes must be positive");
    if (cfg->global_batch_size < cfg->batch_size) SERT_FAIL("global_batch_size < batch_size");
    if (cfg->id_bytes != 1 && cfg->id_bytes != 2 && cfg->id_bytes != 4) SERT_FAIL("id_bytes must be 1, 2 or 4");
    if (cfg->kind != SERT_KIND_LOGLINEAR) {
        if (cfg->entity_dim <= 0 || cfg->entity_dim > 512) SERT_FAIL("entity_dim must be in [1, 512]");
        if (cfg->num_negatives < 0) SERT_FAIL("num_negatives must be >= 0");
    }
    SERT_HIP(hipSetDevice(cfg->device));
    sert_model* m = new sert_model();
    m->cfg = *cfg;
    // everything below may fail half-way (out of memory, ...): the partially built model is
    // torn down by sert_destroy, which tolerates null members
    const int rc = create_resources(m);
    if (rc != 0) {
        const std::string why = g_last_error;   // sert_destroy must not clobber the message
        sert_destroy(m);
        g_last_error = why;
        return rc;
    }
    *out = m;
    return 0;
}

// (Re)build the flat gradient buffer [g_rw | g_re | g_w | g_b | loss sum, owned sum of squares,
// pad | per-entity run bounds] for the current paddings pt_pad[] (every sub-tensor 16-byte aligned).
static int layout_gradients(sert_model* m) {
    const bool vs = is_vs(m);
    const size_t V = m->cfg.num_entities;
    // Order inside the buffer: the word table first (gflat[0, ar_split)), then the other SHARDED
    // tensors, then the replicated ones -- whatever their index: a loglinear model has no R_e and
    // may have a big W (d_w V_e > 4 M), a vectorspace model a small R_e beside a big W -- so that
    // the replicated remainder [rest_off, gflat_count) is one contiguous all-reduce.
    size_t off[5];
    size_t cur = 0;
    auto place = [&](int i) { off[i] = cur; cur += round_up(m->pt_pad[i], 4); };
    place(0);
    m->ar_split = cur;
    for (int i = 1; i < 4; ++i) if (m->pt_sharded[i]) place(i);
    m->rest_off = m->pt_sharded[0] ? cur : 0;
    for (int i = 1; i < 4; ++i) if (!m->pt_sharded[i]) place(i);
    off[4] = cur;
    m->gflat_count = off[4] + 4;
    // tail of the same allocation (zeroed with the gradients every step, not part of any
    // exchange): per-entity sorted-run bounds
    m->gflat_alloc = m->gflat_count + (vs ? 2 * round_up(V, 4) : 0);
    if (m->gflat) { SERT_HIP(hipStreamSynchronize(m->stream)); (void)hipFree(m->gflat); m->gflat = nullptr; }
    SERT_TRY(dzalloc(&m->gflat, m->gflat_alloc, m->stream));
    if (vs) {
        m->run_start = (int32_t*)(m->gflat + m->gflat_count);
        m->run_end = m->run_start + round_up(V, 4);
    }
    m->g_rw = m->gflat + off[0];
    m->g_re = m->n_re ? m->gflat + off[1] : nullptr;
    m->g_w = m->gflat + off[2];
    m->g_b = m->gflat + off[3];
    m->g_loss = m->gflat + off[4];
    m->g_sq = m->g_loss + 1;
    return 0;
}

static int create_resources(sert_model* m) {
    const auto& c = m->cfg;
    SERT_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    SERT_HIP(hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
    // The opt-in schedules that need a third / fourth queue create them; by default they do not exist:
    // HIP maps streams onto FOUR hardware queues, and a fifth stream -- the communication stream of a
    // data-parallel model -- would share one with the main stream (its kernels then queue behind the
    // main stream's, and every cross-stream event costs 10-30 us instead of ~6)
    {
        const char* e3 = knob("SERT_STREAMS");
        const bool want3 = (e3 && atoi(e3) >= 3) || (variant_knob("SERT_DW_THIRD") && atoi(variant_knob("SERT_DW_THIRD")) != 0);
        const bool want4 = variant_knob("SERT_ADAM_SPLIT") && atoi(variant_knob("SERT_ADAM_SPLIT")) != 0;
        if (want3) SERT_HIP(hipStreamCreateWithFlags(&m->stream3, hipStreamNonBlocking));
        if (want4) SERT_HIP(hipStreamCreateWithFlags(&m->stream4, hipStreamNonBlocking));
    }
    // Events that only order this device's own streams against each other need no SYSTEM-scope release (the cache
    // write-back + invalidate that makes device memory visible to the host and to other devices): every kernel ends with
    // an agent-scope release already.  hipEventDisableSystemFence; SERT_EVENT_FENCE=system restores the default flags.
    // (ev_loss is waited on by the HOST and keeps them; so do the events of a communicator, whose consumers may be peers.)
    // (opt-in: measured SLOWER than the two launches at C2 -- 53 us against 25 + 25 -- and equal at 8192 rows; kernels_proj.h)
    m->proj_fused = variant_knob("SERT_PROJ_FUSED") && atoi(variant_knob("SERT_PROJ_FUSED")) != 0;
    m->lazy_skip = !(knob("SERT_LAZY_SKIP") && atoi(knob("SERT_LAZY_SKIP")) == 0);
    m->lazy_max = knob("SERT_LAZY_MAX") ? (float)atof(knob("SERT_LAZY_MAX")) : 0.5f;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->cfg.device) != hipSuccess || cus <= 0) cus = 256;
        m->num_cus = cus;
    }
    const char* fence_env = knob("SERT_EVENT_FENCE");
    SERT_TRY(create_intra_events(m, !(fence_env && !strcmp(fence_env, "system"))));
    SERT_HIP(hipEventCreateWithFlags(&m->ev_loss, hipEventDisableTiming));
    const size_t B = c.batch_size, n = c.window_size, dw = c.word_dim, V = c.num_entities;
    const bool vs = is_vs(m);
    const size_t de = vs ? c.entity_dim : 0;
    m->n_rw = (size_t)c.vocab_size * dw;
    m->n_re = vs ? V * de : 0;
    m->n_w = vs ? dw * de : dw * V;
    m->n_b = vs ? de : V;
    hipStream_t s = m->stream;
    {
        const size_t n[4] = {m->n_rw, m->n_re, m->n_w, m->n_b};
        for (int i = 0; i < 4; ++i) {
            m->pt_pad[i] = n[i];
            // the word table always has its own streaming launch; R_e / W beyond 4 M elements too
            m->pt_big[i] = n[i] > 0 && (i == 0 || (i < 3 && n[i] > ((size_t)1 << 22)));
            m->pt_sharded[i] = false;
        }
    }
    SERT_TRY(dzalloc(&m->rw, m->n_rw, s));  SERT_TRY(dzalloc(&m->re, m->n_re, s));
    SERT_TRY(dzalloc(&m->W, m->n_w, s));    SERT_TRY(dzalloc(&m->b, m->n_b, s));
    if (!c.inference_only) {
        SERT_TRY(dzalloc(&m->s0_rw, m->n_rw, s)); SERT_TRY(dzalloc(&m->s0_re, m->n_re, s));
        SERT_TRY(dzalloc(&m->s0_w, m->n_w, s));   SERT_TRY(dzalloc(&m->s0_b, m->n_b, s));
        SERT_TRY(dzalloc(&m->s1_rw, m->n_rw, s)); SERT_TRY(dzalloc(&m->s1_re, m->n_re, s));
        SERT_TRY(dzalloc(&m->s1_w, m->n_w, s));   SERT_TRY(dzalloc(&m->s1_b, m->n_b, s));
        // the lazy word-table update's per-row update counters (two: the kernel reads one and writes the other)
        static const bool no_lazy = variant_knob("SERT_NO_LAZY") != nullptr;   // cross-check knob (variants build)
        if (!no_lazy && c.word_dim % 4 == 0) {
            SERT_TRY(dzalloc(&m->rw_last[0], (size_t)c.vocab_size, s));
            SERT_TRY(dzalloc(&m->rw_last[1], (size_t)c.vocab_size, s));
            m->rw_pred_stride = (unsigned)round_up((size_t)c.vocab_size, 64);
            SERT_TRY(dzalloc(&m->rw_pred, (size_t)kLazyK * m->rw_pred_stride, s));
        }
        SERT_TRY(layout_gradients(m));
        SERT_TRY(dzalloc(&m->rowloss, B, s));
        size_t part = 0;
        if (vs) {
            SERT_TRY(dzalloc(&m->H, B * dw, s));  SERT_TRY(dzalloc(&m->T, B * de, s));
            if (c.kind == SERT_KIND_VECTORSPACE && !c.inference_only) SERT_TRY(dzalloc(&m->T_alt, B * de, s));
            SERT_TRY(dzalloc(&m->DA, B * de, s)); SERT_TRY(dzalloc(&m->DH, B * dw, s));
            SERT_TRY(dzalloc(&m->neg, std::max<size_t>(4, B * c.num_negatives), s));
            SERT_TRY(dzalloc(&m->neg_alt, std::max<size_t>(4, B * c.num_negatives), s));
            SERT_TRY(dzalloc(&m->neg_stage, std::max<size_t>(4, B * c.num_negatives), s));
            part = (size_t)1024 * (dw * de + de);
            if (c.kind == SERT_KIND_VECTORSPACE_SOFTMAX) {
                // The logit matrix is never larger than ~1.7 GB (SERT_FS_TILE_MB): beyond that the step walks
                // row tiles -- logits, softmax cross-entropy, dR_e += dZ^T.p and dp = dZ.R_e per tile -- so the
                // C4 configuration (65536 x 100 000 logits = 26 GB as one matrix) needs 1.6 GB of scratch.
                // SERT_FS_TILE_ROWS forces a tile height (tests).
                {
                    const char* em = variant_knob("SERT_FS_TILE_MB");
                    const size_t cap = (size_t)(em && atoi(em) > 0 ? atoi(em) : 1700) << 20;
                    size_t tile = B;
                    if (B * V * sizeof(float) > cap) tile = std::max<size_t>(256, (cap / (V * sizeof(float))) / 256 * 256);
                    const char* er = knob("SERT_FS_TILE_ROWS");
                    if (er && atoi(er) > 0) tile = (size_t)atoi(er);
                    m->fs_tile = (int)std::min<size_t>(B, tile);
                }
                SERT_TRY(dzalloc(&m->Z, (size_t)m->fs_tile * V, s));       // logits -> dL/dlogits, one row tile
                SERT_TRY(dzalloc(&m->DH2, B * de, s));    // p = clip(t)
                const size_t tiles = (size_t)cdiv(de, GN) * cdiv(V, GM);
                const size_t sp = std::max<size_t>(1, cdiv(1024, tiles)) + 1;
                part = std::max(part, sp * V * de);
            }
            const size_t total = B * (c.num_negatives + 1);
            {
                // sort-free entity gradient for small vocabularies (kernels_egrad.h); SERT_EGRAD_SORT=1
                // keeps the sorted path (cross-check knob)
                const bool force_sort = knob("SERT_EGRAD_SORT") && atoi(knob("SERT_EGRAD_SORT")) != 0;   // (read per model)
                m->egrad_force_sort = force_sort;
                m->egrad_ranges = variant_knob("SERT_EGRAD_RANGES") && atoi(variant_knob("SERT_EGRAD_RANGES")) != 0;
                const size_t c1 = c.num_negatives + 1;
                if (!force_sort && c.kind == SERT_KIND_VECTORSPACE && V <= 2048 && de % 4 == 0 && de <= 128 &&
                    total < ((size_t)1 << 27) && c1 <= (size_t)kElSubPairs) {
                    m->eg_er_shift = 4;                                      // 16 entities per range (egrad_acc)
                    m->eg_ranges = cdiv(V, 16);                              // <= 128 = kElMaxRanges
                    m->eg_sub_rows = (int)std::min<size_t>(256, kElSubPairs / c1);
                    // small batches: finer sub-groups and row groups, so that the accumulation still
                    // launches ~16 row groups x ranges workgroups with all four waves at work
                    // (batch 4096 was ONE row group: 63 workgroups, 31 us; now 16 x 63)
                    while (m->eg_sub_rows > 32 && (size_t)B < (size_t)64 * m->eg_sub_rows) m->eg_sub_rows /= 2;
                    m->eg_num_sub = cdiv(B, m->eg_sub_rows);
                    // row groups whose slice of T (rows x d_e floats) stays in one XCD's L2: <= 2 MB
                    m->eg_subs_per_group = (int)std::max<size_t>(1, (((size_t)2 << 20) / (de * sizeof(float))) / m->eg_sub_rows);
                    static const int want_groups = variant_knob("SERT_EG_GROUPS") ? std::max(1, atoi(variant_knob("SERT_EG_GROUPS"))) : 16;   // tuning knob
                    m->eg_subs_per_group = std::max(1, std::min(m->eg_subs_per_group, m->eg_num_sub / want_groups));
                    m->eg_groups = cdiv(m->eg_num_sub, m->eg_subs_per_group);
                }
                if (m->eg_groups > 0) {
                    SERT_TRY(dzalloc(&m->epart, (size_t)m->eg_groups * V * de, s));
                    SERT_TRY(dzalloc(&m->eg_entries, total, s));
                    SERT_TRY(dzalloc(&m->eg_offs, (size_t)m->eg_num_sub * (m->eg_ranges + 1), s));
                }
            }
            SERT_TRY(dzalloc(&m->cand, total, s));        SERT_TRY(dzalloc(&m->cand_sorted, total + 1, s));
            SERT_TRY(dzalloc(&m->cand_early, total, s));
            SERT_TRY(dzalloc(&m->pair_sorted, total, s));
            SERT_TRY(dzalloc(&m->coef, total, s));
            const size_t chunks = (total + kEChunk - 1) / kEChunk;
            SERT_TRY(dzalloc(&m->ehead, chunks * de, s)); SERT_TRY(dzalloc(&m->etail, chunks * de, s));
            m->sort_bits = 1;
            while ((1ll << m->sort_bits) < (long long)V) ++m->sort_bits;
            const size_t tiles = (total + kSortTile - 1) / kSortTile;
            SERT_TRY(dzalloc(&m->sort_hist, (size_t)kSortMaxBins * tiles, s));
            SERT_TRY(dzalloc(&m->sort_bin_total, (size_t)kSortMaxBins, s));
            if (m->sort_bits > kSortMaxBits) {
                SERT_TRY(dzalloc(&m->sort_k_tmp, total, s));
                SERT_TRY(dzalloc(&m->sort_v_tmp, total, s));
            }
        } else {
            // the fused loss kernel may ask for more than the default 64 KB of dynamic LDS
#define SERT_LL_ATTR(NT)                                                                                  \
    SERT_HIP(hipFuncSetAttribute((const void*)ll_fused_row<true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
    SERT_HIP(hipFuncSetAttribute((const void*)ll_fused_row<false, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
            SERT_LL_ATTR(512);
            SERT_HIP(hipFuncSetAttribute((const void*)ll_row_from_table<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#undef SERT_LL_ATTR
            SERT_TRY(dzalloc(&m->G, B * n * dw, s));  SERT_TRY(dzalloc(&m->Z, B * n * V, s));
            SERT_TRY(dzalloc(&m->J, B * V, s));       SERT_TRY(dzalloc(&m->DG, B * n * dw, s));
            const size_t nseg = cdiv(V, kLlSeg);
            SERT_TRY(dzalloc(&m->ll_tokstat, B * n * nseg, s)); SERT_TRY(dzalloc(&m->ll_lse, B * n, s));
            SERT_TRY(dzalloc(&m->ll_jstat, B * nseg, s));       SERT_TRY(dzalloc(&m->ll_rowinfo, B, s));
            SERT_TRY(dzalloc(&m->ll_rpart, B * n * nseg, s));   SERT_TRY(dzalloc(&m->ll_r, B * n, s));
            SERT_TRY(dzalloc(&m->ll_rsum, B * n, s));
            const size_t tiles = (size_t)cdiv(V, GN) * cdiv(dw, GM);
            const size_t splits = std::max<size_t>(1, cdiv(1024, tiles)) + 1;
            part = splits * (dw * V + V);
        }
        m->part_count = part;
        SERT_TRY(dzalloc(&m->part, part, s));
        SERT_TRY(dzalloc(&m->red_loss, std::max<size_t>((size_t)kOptBlocks, (B + 15) / 16), s));
        SERT_TRY(dzalloc(&m->red_sq, (size_t)8 * kOptBlocks, s));  // partials of up to 4 tensors (<= 2 kOptBlocks each)
        SERT_TRY(dzalloc(&m->re_sq, (size_t)4 * kOptBlocks, s));
        SERT_TRY(dzalloc(&m->sq_scratch, (size_t)8 * kOptBlocks, s));
        SERT_TRY(dzalloc(&m->d_loss, (size_t)4, s));
        if (vs) {
            SERT_TRY(dzalloc(&m->tail_blk, (size_t)4 * (cdiv((int64_t)(m->n_w + m->n_b), 64) + 1), s));
        }
    }
    // pinned, device-mapped: [loss, data, reg, -, seq]; the step's last kernel writes it directly
    SERT_HIP(hipHostMalloc((void**)&m->h_loss, 8 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    memset(m->h_loss, 0, 8 * sizeof(float));
    SERT_HIP(hipHostGetDevicePointer((void**)&m->h_loss_dev, m->h_loss, 0));
    {
        const char* e = knob("SERT_STREAMS");   // tuning / cross-check knob
        const int v = e ? atoi(e) : 0;
        if (v >= 1 && v <= 3) m->nstreams = v;
    }
    for (int g = 0; g < TG_COUNT; ++g)
        for (int k = 0; k < 2; ++k) SERT_HIP(hipEventCreate(&m->timing.ev[g][k]));
    m->timing.created = true;
    SERT_HIP(hipStreamSynchronize(s));
    return 0;
}

static void free_split(DataSplit& d) {
    (void)hipFree(d.x); (void)hipFree(d.y); (void)hipFree(d.csr_indptr);
    (void)hipFree(d.csr_indices); (void)hipFree(d.csr_data); (void)hipFree(d.w); (void)hipFree(d.labfix);
    (void)hipFree(d.idx_rows); (void)hipFree(d.idx_items);
    (void)hipFree(d.idx_heavy); d.idx_heavy = nullptr;
    (void)hipFree(d.idx_bundles); d.idx_bundles = nullptr;
    (void)hipFree(d.idx_uwords); (void)hipFree(d.idx_slots); (void)hipFree(d.idx_rows_div);
    (void)hipFree(d.idx_touched_bits);
    (void)hipFree(d.idx_dense_counts); (void)hipFree(d.idx_dense_words); (void)hipFree(d.idx_tok_slot);
    d.idx_uwords = nullptr; d.idx_slots = nullptr; d.idx_rows_div = nullptr;
    d = DataSplit();
}

int sert_destroy(sert_model* m) {
    if (!m) return 0;
    (void)hipSetDevice(m->cfg.device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->stream2) (void)hipStreamSynchronize(m->stream2);   // (work that ran ahead of the host)
    if (m->stream3) (void)hipStreamSynchronize(m->stream3);
    if (m->stream4) (void)hipStreamSynchronize(m->stream4);
    if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
    if (m->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
    if (m->ev_rest_ready) (void)hipEventDestroy(m->ev_rest_ready);
    if (m->ev_ar_done) (void)hipEventDestroy(m->ev_ar_done);
    if (m->ev_ag_done) (void)hipEventDestroy(m->ev_ag_done);
    for (int i = 0; i < 4; ++i) {
        if (m->ev_grad_ready[i]) (void)hipEventDestroy(m->ev_grad_ready[i]);
        for (int c = 0; c < sert_model::kMaxArChunks; ++c) {
            if (m->ev_rs_done[i][c]) (void)hipEventDestroy(m->ev_rs_done[i][c]);
            if (m->ev_opt_done[i][c]) (void)hipEventDestroy(m->ev_opt_done[i][c]);
        }
    }
    (void)hipFree(m->sq_scratch);
    (void)hipFree(m->tail_blk);
    (void)hipFree(m->rw_last[0]); (void)hipFree(m->rw_last[1]); (void)hipFree(m->rw_pred);
    xr_free_lists(m);
    if (m->ev_params_ready) (void)hipEventDestroy(m->ev_params_ready);
    if (m->ev_word_updated) (void)hipEventDestroy(m->ev_word_updated);
    if (m->comm_stream) (void)hipStreamDestroy(m->comm_stream);
    float* bufs[] = {m->rw, m->re, m->W, m->b, m->s0_rw, m->s0_re, m->s0_w, m->s0_b, m->s1_rw,
                     m->s1_re, m->s1_w, m->s1_b, m->gflat, m->H, m->T, m->T_alt, m->DA, m->DH, m->rowloss,
                     m->G, m->Z, m->J, m->DG, m->DH2, m->part, m->skbuf, m->wpart, m->hpart, m->red_loss, m->red_sq, m->re_sq, m->d_loss,
                     m->d_losses};
    for (float* p : bufs) (void)hipFree(p);
    (void)hipFree(m->ll_tokstat); (void)hipFree(m->ll_lse); (void)hipFree(m->ll_jstat);
    (void)hipFree(m->ll_rowinfo); (void)hipFree(m->ll_rpart); (void)hipFree(m->ll_r); (void)hipFree(m->ll_rsum);
    (void)hipFree(m->Zu); (void)hipFree(m->dZu); (void)hipFree(m->zpart);
    (void)hipFree(m->neg); (void)hipFree(m->neg_alt); (void)hipFree(m->neg_stage);
    (void)hipFree(m->pred_a); (void)hipFree(m->pred_b); (void)hipFree(m->pred_ids);
    (void)hipFree(m->cand); (void)hipFree(m->cand_sorted); (void)hipFree(m->cand_early);
    (void)hipFree(m->pair_sorted); (void)hipFree(m->coef); (void)hipFree(m->ehead);
    (void)hipFree(m->etail); (void)hipFree(m->epart); (void)hipFree(m->eg_entries); (void)hipFree(m->eg_offs); (void)hipFree(m->sort_hist); (void)hipFree(m->sort_bin_total);
    (void)hipFree(m->sort_k_tmp); (void)hipFree(m->sort_v_tmp);
    if (m->h_loss) (void)hipHostFree(m->h_loss);
    if (m->host_send) (void)hipHostFree(m->host_send);
    if (m->host_recv) (void)hipHostFree(m->host_recv);
    free_split(m->split[0]); free_split(m->split[1]);
    if (m->timing.created)
        for (int g = 0; g < TG_COUNT; ++g)
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(m->timing.ev[g][k]);
    if (m->instep.created)
        for (int i = 0; i < InStep::kRing; ++i)
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(m->instep.ev[i][k]);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->stream2) (void)hipStreamDestroy(m->stream2);
    if (m->ev_join3) (void)hipEventDestroy(m->ev_join3);
    for (hipEvent_t e : {m->ev_step_done, m->ev_neg, m->ev_opt_fork, m->ev_small, m->ev_re, m->ev_dense, m->ev_loss})
        if (e) (void)hipEventDestroy(e);
    if (m->stream3) (void)hipStreamDestroy(m->stream3);
    if (m->stream4) (void)hipStreamDestroy(m->stream4);
    for (hipEvent_t e : {m->ev_word_opt, m->ev_early})
        if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return 0;
}

size_t sert_tensor_size(sert_model* m, int which) {
    if (!m) return 0;
    return tensor_ref(m, which).count;
}

int sert_set_tensor(sert_model* m, int which, const float* host, size_t count) {
    if (m) invalidate_speculation(m);
    if (!m || !host) SERT_FAIL("null argument");
    SERT_HIP(hipSetDevice(m->cfg.device));
    TensorRef t = tensor_ref(m, which);
    if (!t.ptr || t.count == 0) SERT_FAIL("tensor not present for this model kind");
    if (t.count != count) SERT_FAIL("element count mismatch");
    if (which >= SERT_T_STATE0_RW && which <= SERT_T_STATE1_B && m->pt_sharded[(which - SERT_T_STATE0_RW) % 4])
        return sharded_state_io(m, (which - SERT_T_STATE0_RW) % 4, (which - SERT_T_STATE0_RW) / 4, nullptr, host);
    if (m->comm_stream) SERT_HIP(hipStreamSynchronize(m->comm_stream));
    SERT_TRY(settle_entity_update(m));
    m->re_sq_for[0] = m->re_sq_for[1] = -1;
    SERT_HIP(hipMemcpyAsync(t.ptr, host, count * sizeof(float), hipMemcpyHostToDevice, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    if (which == SERT_T_RW) { m->rw_full = true; m->xr_fetched_batch = -1; }   // (every rank sets the whole table)
    return 0;
}

int sert_get_tensor(sert_model* m, int which, float* host, size_t count) {
    if (!m || !host) SERT_FAIL("null argument");
    SERT_HIP(hipSetDevice(m->cfg.device));
    TensorRef t = tensor_ref(m, which);
    if (!t.ptr || t.count == 0) SERT_FAIL("tensor not present for this model kind");
    if (t.count != count) SERT_FAIL("element count mismatch");
    // Without keep_grads the gradient buffers are scratch: the word table is neither zeroed
    // nor fully written, the L2 term is never stored, and the activation buffers may already
    // hold the NEXT batch (sert_hint_next_batch).  Refuse instead of returning something stale.
    if (which >= SERT_T_GRAD_RW && which <= SERT_T_ACT_ROWLOSS && !m->cfg.keep_grads)
        SERT_FAIL("gradients and activations are only readable from a model created with keep_grads = 1");
    if (which >= SERT_T_STATE0_RW && which <= SERT_T_STATE1_B && m->pt_sharded[(which - SERT_T_STATE0_RW) % 4])
        return sharded_state_io(m, (which - SERT_T_STATE0_RW) % 4, (which - SERT_T_STATE0_RW) / 4, host, nullptr);
    if (which == SERT_T_RW) SERT_TRY(ensure_full_rw(m));   // (owned by rows: collective while stale)
    SERT_TRY(ensure_rw_current(m, -1));                     // (lazy word-table update: flush before anyone looks)
    if (m->comm_stream) SERT_HIP(hipStreamSynchronize(m->comm_stream));
    if (m->stream2) SERT_HIP(hipStreamSynchronize(m->stream2));
    SERT_HIP(hipMemcpyAsync(host, t.ptr, count * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    return 0;
}

int sert_set_step(sert_model* m, int64_t t) {
    if (!m || t < 0) SERT_FAIL("bad argument");
    invalidate_speculation(m);     // (first: a lazy word table is flushed at the OLD step count)
    m->step = t;
    return 0;
}
int64_t sert_get_step(sert_model* m) { return m ? m->step : -1; }

int sert_set_eval_draws(sert_model* m, int64_t n) {
    if (!m || n < 0) SERT_FAIL("bad argument");
    m->eval_draws = n;
    return 0;
}
int64_t sert_get_eval_draws(sert_model* m) { return m ? m->eval_draws : -1; }

int sert_negatives_of_step(sert_model* m, int64_t position, int evaluation, int64_t* out) {
    if (!m || !out || position < 0) SERT_FAIL("bad argument");
    if (!is_vs(m) || is_fs(m) || m->cfg.num_negatives <= 0) SERT_FAIL("this model draws no negatives");
    SERT_HIP(hipSetDevice(m->cfg.device));
    const int64_t count = (int64_t)m->cfg.batch_size * m->cfg.num_negatives;
    int32_t* tmp = nullptr;
    SERT_HIP(hipMalloc((void**)&tmp, (size_t)count * sizeof(int32_t)));
    // a stream of its own: nothing of the model's state or schedule is touched
    hipStream_t st = nullptr;
    if (hipStreamCreate(&st) != hipSuccess) { (void)hipFree(tmp); SERT_FAIL("hipStreamCreate failed"); }
    hipLaunchKernelGGL(vs_sample_negatives, dim3(grid_for((count + 3) / 4)), dim3(256), 0, st, tmp, count,
                       (int64_t)m->rank * count, (uint32_t)m->cfg.num_entities, m->cfg.seed,
                       (uint64_t)position * 2 + (evaluation ? 1 : 0), (float4*)nullptr, (size_t)0, (uint4*)nullptr, (size_t)0);
    std::vector<int32_t> host((size_t)count);
    const hipError_t e = hipMemcpyAsync(host.data(), tmp, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    (void)hipFree(tmp);
    if (e != hipSuccess || e2 != hipSuccess) SERT_FAIL("reading the negatives back failed");
    for (int64_t i = 0; i < count; ++i) out[i] = host[(size_t)i];
    return 0;
}

int sert_upload_dataset(sert_model* m, int split, const void* x, const int32_t* y_int,
                        const int64_t* csr_indptr, const int32_t* csr_indices,
                        const float* csr_data, const float* w, int64_t N) {
    if (!m) SERT_FAIL("null model");
    invalidate_speculation(m);
    m->hint_next = -1;
    if (split != SERT_SPLIT_TRAIN && split != SERT_SPLIT_VALIDATE) SERT_FAIL("bad split");
    if (N < 0) SERT_FAIL("negative instance count");
    if (N > 0 && !x) SERT_FAIL("x is null");
    if (N > 0 && !y_int && !(csr_indptr && csr_indices && csr_data)) SERT_FAIL("no labels given");
    if (is_vs(m) && N > 0 && !y_int) SERT_FAIL("vectorspace requires int labels (models.py:933-934)");
    SERT_HIP(hipSetDevice(m->cfg.device));
    // Every id the kernels will index with is checked here, once, on the host: the reference
    // raises IndexError for an out-of-range token or label (numpy / Theano advanced indexing);
    // on the device it would be an out-of-bounds read, and in the backward a write.
    if (N > 0) {
        const size_t toks = (size_t)N * m->cfg.window_size;
        const uint32_t Vw = (uint32_t)m->cfg.vocab_size;
        bool ok = true;
        SERT_ID_DISPATCH(m->cfg.id_bytes, {
            const IdT* xi = (const IdT*)x;
            uint32_t mx = 0;
            for (size_t i = 0; i < toks; ++i) mx = std::max<uint32_t>(mx, (uint32_t)xi[i]);
            ok = mx < Vw;
        });
        if (!ok) SERT_FAIL("token id >= vocab_size in x");
        const int32_t Ve = m->cfg.num_entities;
        if (y_int) {
            for (int64_t i = 0; i < N; ++i)
                if (y_int[i] < 0 || y_int[i] >= Ve) SERT_FAIL("label out of range [0, num_entities) in y");
        } else {
            if (csr_indptr[0] != 0) SERT_FAIL("csr_indptr[0] != 0");
            for (int64_t r = 0; r < N; ++r)
                if (csr_indptr[r + 1] < csr_indptr[r]) SERT_FAIL("csr_indptr is not non-decreasing");
            const int64_t nnz = csr_indptr[N];
            for (int64_t i = 0; i < nnz; ++i)
                if (csr_indices[i] < 0 || csr_indices[i] >= Ve) SERT_FAIL("label column out of range [0, num_entities) in csr_indices");
        }
    }
    DataSplit& d = m->split[split];
    free_split(d);
    d.N = N;
    if (N == 0) return 0;
    hipStream_t s = m->stream;
    const size_t xbytes = (size_t)N * m->cfg.window_size * m->cfg.id_bytes;
    SERT_HIP(hipMalloc(&d.x, xbytes));
    SERT_HIP(hipMemcpyAsync(d.x, x, xbytes, hipMemcpyHostToDevice, s));
    if (y_int) {
        d.max_labels_per_row = 1;
        SERT_TRY(dmalloc(&d.y, (size_t)N));
        SERT_HIP(hipMemcpyAsync(d.y, y_int, N * sizeof(int32_t), hipMemcpyHostToDevice, s));
    } else {
        d.nnz = csr_indptr[N];
        for (int64_t r = 0; r < N; ++r)
            d.max_labels_per_row = std::max<int64_t>(d.max_labels_per_row, csr_indptr[r + 1] - csr_indptr[r]);
        SERT_TRY(dmalloc(&d.csr_indptr, (size_t)N + 1));
        SERT_TRY(dmalloc(&d.csr_indices, (size_t)std::max<int64_t>(1, d.nnz)));
        SERT_TRY(dmalloc(&d.csr_data, (size_t)std::max<int64_t>(1, d.nnz)));
        SERT_HIP(hipMemcpyAsync(d.csr_indptr, csr_indptr, (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        if (d.nnz) {
            SERT_HIP(hipMemcpyAsync(d.csr_indices, csr_indices, d.nnz * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_HIP(hipMemcpyAsync(d.csr_data, csr_data, d.nnz * sizeof(float), hipMemcpyHostToDevice, s));
        }
    }
    if (split == SERT_SPLIT_TRAIN) {
        SERT_TRY(dmalloc(&d.w, (size_t)N));
        if (w) {
            SERT_HIP(hipMemcpyAsync(d.w, w, N * sizeof(float), hipMemcpyHostToDevice, s));
        } else {
            std::vector<float> ones((size_t)N, 1.0f);
            SERT_HIP(hipMemcpyAsync(d.w, ones.data(), N * sizeof(float), hipMemcpyHostToDevice, s));
            SERT_HIP(hipStreamSynchronize(s));
        }
    }
    if (split == SERT_SPLIT_TRAIN && !m->cfg.inference_only && !is_vs(m))
        SERT_TRY(dmalloc(&d.labfix, (size_t)std::max<int64_t>(1, y_int ? N : d.nnz)));
    if (split == SERT_SPLIT_TRAIN && !m->cfg.inference_only) {
        const int B = m->cfg.batch_size, n = m->cfg.window_size;
        const int64_t nb = N / B;
        WordIndex wi;
        const bool row_is_pos = !is_vs(m);
        // vectorspace models: the heavy words of a batch are summed by one dense pass (word_index.h);
        // SERT_NO_DENSE_HEAVY=1 keeps them in the tree (cross-check knob)
        // (loglinear: the V_e-wide per-word sums are bandwidth-bound; on by default where V_e % 4 == 0.  vectorspace: as two
        //  launches in FRONT of the tree the pass cost what the heavy words' entries saved -- round 3: 56.4 against 55.6 us at C2,
        //  opt-in then.  Round 5: inside the tree's own launches (kernels_seg.h: segsum_rows_plus -- the stream beside the
        //  latency-bound level 0, the combine beside level 1) the C2 tree takes 42.6 us instead of 55.8 and the step 0.2481 ->
        //  0.2374 ms, C4 1.386 -> 1.361, C2 dims at 8192 rows 0.0988 -> 0.0964, product-search settings 0.1762 -> 0.1750
        //  (tools/experiments/r05_heavy_fused.sh; the two launches in front: 0.257, 1.440, 0.114, 0.181).  ON by default where
        //  the tree runs its 32-lane forms (d_w / 4 <= 32, or rows that three 32-lane column groups cover better than two
        //  64-lane ones); SERT_DENSE_HEAVY=0 / 1 forces it off / on.)
        bool vs_heavy = false;
        if (is_vs(m) && m->cfg.word_dim % 4 == 0 && m->cfg.word_dim <= 512) {
            const int d4 = m->cfg.word_dim / 4;
            const bool lpi32 = d4 <= 32 || (d4 > 64 && 64 * cdiv(d4, 64) > 32 * cdiv(d4, 32));
            const bool bundle_on = variant_knob("SERT_SEG_BUNDLE") && atoi(variant_knob("SERT_SEG_BUNDLE")) != 0;
            vs_heavy = knob("SERT_DENSE_HEAVY") ? atoi(knob("SERT_DENSE_HEAVY")) != 0 : (lpi32 && !bundle_on);
        }
        const bool dense_heavy = !variant_knob("SERT_NO_DENSE_HEAVY") && (is_vs(m) ? vs_heavy : (m->cfg.num_entities % 4 == 0));
        // Row-grouped level 0 of the vectorspace word-gradient tree (word_index.h: row_groups; kernels_seg.h: XcdLists):
        // MEASURED AND NOT USED (round 4, profiles/r04_experiments.txt).  At C2 it does what it was built for -- the
        // fabric traffic of the tree falls from 264 MB to 149 MB per step (level 0: 237 -> 98 MB) -- and level 0 takes
        // the same 41.5 us while the upper level grows from 8 to 14 us (100 k items and 75 k partial rows instead of
        // 52 k and 10 k): the step 0.2945 -> 0.3114 ms with 8 row ranges, 0.318 with 16, 0.325 with 32.
        // SERT_SEG_GROUPS=k builds it (tests/test_gpu_parity.py::test_word_gradient_row_grouped_tree keeps it exact).
        int row_groups = 1;
        if (is_vs(m) && m->cfg.word_dim % 4 == 0)
            if (const char* e = variant_knob("SERT_SEG_GROUPS")) row_groups = std::min(std::max(1, atoi(e)), std::max(1, B / 64));
        // vectorspace: level 0 sorted by item length with the first row number in the descriptor (word_index.h: slot_is_row);
        // not with bundles (they need the items in entry order); SERT_SEG_NO_SORT (variants build) for the A/B
        const bool sort_level0 = is_vs(m) && row_groups == 1 && !(variant_knob("SERT_SEG_BUNDLE") && atoi(variant_knob("SERT_SEG_BUNDLE")) != 0) &&
                                 !variant_knob("SERT_SEG_NO_SORT");
        bool ids_ok = true;
        SERT_ID_DISPATCH(m->cfg.id_bytes,
                         ids_ok = build_word_index<IdT>((const IdT*)x, nb, B, n, m->cfg.vocab_size, row_is_pos, wi,
                                                        /*want_slots=*/!is_vs(m), /*dense_heavy=*/dense_heavy, row_groups, sort_level0));
        if (!ids_ok) SERT_FAIL("token id >= vocab_size in x");
        if (!is_vs(m) && !wi.slots.empty()) {
            const size_t V = (size_t)m->cfg.num_entities;
            SERT_TRY(dmalloc(&d.idx_uwords, std::max<size_t>(1, wi.uwords.size())));
            SERT_TRY(dmalloc(&d.idx_slots, wi.slots.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_uwords, wi.uwords.data(), wi.uwords.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_HIP(hipMemcpyAsync(d.idx_slots, wi.slots.data(), wi.slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_TRY(dmalloc(&d.idx_rows_div, wi.rows_div.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_rows_div, wi.rows_div.data(), wi.rows_div.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_HIP(hipStreamSynchronize(s));
            if ((size_t)wi.max_distinct > m->zu_rows) {
                (void)hipFree(m->Zu); (void)hipFree(m->dZu);
                m->Zu = nullptr; m->dZu = nullptr;
                m->zu_rows = (size_t)wi.max_distinct;
                SERT_TRY(dmalloc(&m->Zu, m->zu_rows * V));
                SERT_TRY(dmalloc(&m->dZu, m->zu_rows * V));
            }
            if ((size_t)wi.max_part_rows + 1 > m->zpart_rows) {
                (void)hipFree(m->zpart);
                m->zpart = nullptr;
                m->zpart_rows = (size_t)wi.max_part_rows + 1;
                SERT_TRY(dmalloc(&m->zpart, m->zpart_rows * V));
            }
        }
        if (!wi.rows.empty()) {
            SERT_TRY(dmalloc(&d.idx_rows, wi.rows.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_rows, wi.rows.data(), wi.rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_HIP(hipMalloc((void**)&d.idx_items, std::max<size_t>(1, wi.items.size()) * sizeof(SegItem)));
            SERT_HIP(hipMemcpyAsync(d.idx_items, wi.items.data(), wi.items.size() * sizeof(SegItem), hipMemcpyHostToDevice, s));
            if (!wi.heavy.empty()) {
                SERT_HIP(hipMalloc((void**)&d.idx_heavy, wi.heavy.size() * sizeof(int32_t)));
                SERT_HIP(hipMemcpyAsync(d.idx_heavy, wi.heavy.data(), wi.heavy.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            }
            // OPT-IN (SERT_SEG_BUNDLE=1, read at upload: two models of one process can run either way).  Measured SLOWER
            // (round 5, profiles/r05_experiments.txt: C2 word-gradient group 59.5 us against 56.7, C4 150.9 against 145.0): the
            // short items are not what level 0 waits for -- with their lane groups leaving right after the descriptor load
            // (a knock-out) the group loses 7 of 57 us.
            const bool bundle = variant_knob("SERT_SEG_BUNDLE") && atoi(variant_knob("SERT_SEG_BUNDLE")) != 0;
            if (!wi.bundles.empty() && bundle && is_vs(m)) {
                SERT_TRY(dmalloc(&d.idx_bundles, wi.bundles.size()));
                SERT_HIP(hipMemcpyAsync(d.idx_bundles, wi.bundles.data(), wi.bundles.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            }
            SERT_HIP(hipStreamSynchronize(s));
        }
        if (!wi.touched_bits.empty()) {
            SERT_TRY(dmalloc(&d.idx_touched_bits, wi.touched_bits.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_touched_bits, wi.touched_bits.data(), wi.touched_bits.size() * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, s));
            SERT_HIP(hipStreamSynchronize(s));
            d.bit_words = wi.bit_words;
        }
        // data parallel, word table owned by rows: the per-batch exchange lists (collective)
        if (is_dp(m)) SERT_TRY(xr_build_lists(m, wi.touched_bits, nb, wi.bit_words));
        if (wi.any_dense) {
            SERT_HIP(hipMalloc((void**)&d.idx_dense_counts, wi.dense_counts.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_dense_counts, wi.dense_counts.data(), wi.dense_counts.size(), hipMemcpyHostToDevice, s));
            std::vector<int32_t> hw((size_t)nb * kHeavyMax, 0);
            for (int64_t b = 0; b < nb; ++b)
                for (int h = 0; h < wi.batches[(size_t)b].dense_cnt; ++h) hw[(size_t)b * kHeavyMax + h] = wi.batches[(size_t)b].dense_word[h];
            // (only where the opt-in gather that reads it is switched on: a byte per token of the data set)
            if (!wi.dense_tok_slot.empty() && variant_knob("SERT_GATHER_HOT") && atoi(variant_knob("SERT_GATHER_HOT")) != 0) {
                SERT_HIP(hipMalloc((void**)&d.idx_tok_slot, wi.dense_tok_slot.size()));
                SERT_HIP(hipMemcpyAsync(d.idx_tok_slot, wi.dense_tok_slot.data(), wi.dense_tok_slot.size(), hipMemcpyHostToDevice, s));
                d.dense_cnt_of.resize((size_t)nb);
                for (int64_t b = 0; b < nb; ++b) d.dense_cnt_of[(size_t)b] = wi.batches[(size_t)b].dense_cnt;
            }
            SERT_TRY(dmalloc(&d.idx_dense_words, hw.size()));
            SERT_HIP(hipMemcpyAsync(d.idx_dense_words, hw.data(), hw.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
            SERT_HIP(hipStreamSynchronize(s));
            if (!m->hpart)
                // (row blocks of the form that runs: the fused stream's heavy_rows_fused(B) rows or the two launches'
                //  kHeavyRowsPerBlock -- sized for the smallest block, 64 rows, a loglinear model at batch 65536 and 100 k
                //  entities asked for 6.5 GB where 1.6 GB is used)
                SERT_TRY(dmalloc(&m->hpart, (size_t)std::max(cdiv(B, heavy_rows_fused(B)), cdiv(B, kHeavyRowsPerBlock)) * kHeavyMax *
                                               (size_t)(is_vs(m) ? m->cfg.word_dim : m->cfg.num_entities)));
        } else {
            for (auto& bxx : wi.batches) bxx.dense_cnt = 0;
        }
        d.idx_batches = wi.batches;
        if ((size_t)wi.max_part_rows + 1 > m->wpart_rows) {
            (void)hipFree(m->wpart);
            m->wpart_rows = (size_t)wi.max_part_rows + 1;
            SERT_TRY(dmalloc(&m->wpart, m->wpart_rows * m->cfg.word_dim));
        }
    }
    SERT_HIP(hipStreamSynchronize(s));
    return 0;
}

int sert_train_batch(sert_model* m, int64_t batch_index, const int64_t* negatives, float* loss_out) {
    refresh_gemm_choice();
    if (!m) SERT_FAIL("null model");
    SERT_HIP(hipSetDevice(m->cfg.device));
    static const bool no_spin = variant_knob("SERT_NO_SPIN") != nullptr;   // cross-check knob
    // sert_hint_next_batch: the next batch's parameter-only forward part goes out behind
    // this step, before the host starts waiting for this step's loss
    const int64_t hint = m->hint_next;
    m->hint_next = -1;
    m->lazy_next = hint;
    auto prefetch_next = [&]() -> int {
        const DataSplit& ds = m->split[SERT_SPLIT_TRAIN];
        // (keep_grads: the caller may read this batch's activations after the call)
        if (hint < 0 || m->timing.enabled || m->cfg.keep_grads) return 0;
        if ((hint + 1) * (int64_t)m->cfg.batch_size > ds.N) return 0;
        if (can_speculate_step(m)) {
            // the whole forward + backward of the announced batch runs ahead: it depends on the
            // parameters (final: this step's update is already in the stream), the data and the
            // step counter only, and writes activations / gradient scratch only.  The UPDATE of
            // that step is not issued before the host has seen this step's loss.
            bool fused = false;
            SERT_TRY(step_forward_backward(m, ds, hint, nullptr, &fused));
            m->spec_fb_batch = hint;
            m->spec_fb_step = m->step;
        } else if (is_vs(m) && !is_fs(m)) {
            // data parallel: the parameter-only part (by rows: behind the fetch of the rows it reads)
            if (m->xr_on) SERT_TRY(xr_fetch_params(m, hint));
            SERT_TRY(vs_project(m, ds, hint));
            m->projected_batch = hint;
        } else if (m->xr_on) {
            SERT_TRY(xr_fetch_params(m, hint));
        }
        return 0;
    };
    if (m->timing.enabled || no_spin) {
        SERT_TRY(train_step_async(m, batch_index, negatives, m->d_loss));
        SERT_HIP(hipMemcpyAsync(m->h_loss, m->d_loss, 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        SERT_HIP(hipEventRecord(m->ev_loss, m->stream));
        SERT_TRY(prefetch_next());
        SERT_HIP(hipEventSynchronize(m->ev_loss));   // (not the stream: the next batch may be running ahead)
        timing_collect(m);
        if (loss_out) *loss_out = m->h_loss[0];
        return 0;
    }
    // The step's last kernel writes the loss straight into pinned host memory followed by
    // a sequence number; the host spins on that word -- no copy kernel and no stream
    // synchronisation on the per-step read-back the reference's epoch loop performs
    // (sert/models.py:369-379): 0.412 -> 0.396 ms/step at C2.  Everything the step did is
    // stream-ordered before that kernel, so the parameters are final when the number appears.
    SERT_TRY(train_step_async(m, batch_index, negatives, m->h_loss_dev, true));
    SERT_TRY(prefetch_next());
    const unsigned want = m->loss_seq;
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(m->h_loss + 4);
    // A faulted step never publishes: ask the stream -- but RARELY.  hipStreamQuery on a stream whose last command carries no
    // completion signal makes the runtime enqueue a marker (a barrier packet) to get one; asked every 16 k spins (~5 us, the
    // flag is a cached read) the first marker landed right behind the run-ahead backward of the NEXT batch, and the next call's
    // word-table update started 6-7 us late behind it at every batch size (round 5: HIP API trace, tools/experiments/
    // r05_hip_trace.sh; sert_train_batches, which never asks, has no such gap).  Now: the clock every 4 k spins, the stream
    // only after 50 ms without a loss and every 50 ms from then on.
    {
        const auto t_spin0 = std::chrono::steady_clock::now();
        double next_query_ms = 50.0;
        for (unsigned spins = 1; *flag != want; ++spins) {
            if ((spins & 0xfff) != 0) continue;
            const double waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_spin0).count();
            if (waited_ms < next_query_ms) continue;
            next_query_ms = waited_ms + 50.0;
            const hipError_t q = hipStreamQuery(m->stream);
            if (q == hipSuccess) {
                if (*flag == want) break;
                SERT_FAIL("training step completed without publishing its loss");
            }
            if (q != hipErrorNotReady) SERT_HIP(q);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (loss_out) *loss_out = m->h_loss[0];
    return 0;
}

int sert_hint_next_batch(sert_model* m, int64_t next_batch_index) {
    if (!m) SERT_FAIL("null model");
    m->hint_next = next_batch_index;
    return 0;
}

int sert_train_batches(sert_model* m, const int64_t* batch_indices, int64_t count, float* losses_out) {
    refresh_gemm_choice();
    if (!m || !batch_indices || count < 0) SERT_FAIL("bad argument");
    m->hint_next = -1;
    SERT_HIP(hipSetDevice(m->cfg.device));
    if (count == 0) return 0;
    if (m->d_losses_cap < count) {
        (void)hipFree(m->d_losses);
        SERT_TRY(dmalloc(&m->d_losses, (size_t)count * 3));
        m->d_losses_cap = count;
    }
    static const bool report_host = variant_knob("SERT_DEBUG_HOST") != nullptr;
    const auto t_host0 = std::chrono::steady_clock::now();
    for (int64_t i = 0; i < count; ++i) {
        m->lazy_next = i + 1 < count ? batch_indices[i + 1] : -1;
        SERT_TRY(train_step_async(m, batch_indices[i], nullptr, m->d_losses + 3 * i));
        if (m->timing.enabled) {  // events are single-slot: drain per step when timing
            SERT_HIP(hipStreamSynchronize(m->stream));
            timing_collect(m);
        }
    }
    if (report_host) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count();
        fprintf(stderr, "[sert] host enqueue: %.1f us/step over %lld steps\n", us / (double)count, (long long)count);
    }
    std::vector<float> tmp((size_t)count * 3);
    SERT_HIP(hipMemcpyAsync(tmp.data(), m->d_losses, count * 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    if (losses_out)
        for (int64_t i = 0; i < count; ++i) losses_out[i] = tmp[3 * i];
    return 0;
}

// One evaluation pass over a batch, enqueued on the main stream: forward + loss kernel, then the
// unweighted, unregularised batch mean (models.py:751-752) into dst[0] (device).
static int eval_step_async(sert_model* m, const DataSplit& ds, int64_t batch_index, const int64_t* negatives,
                           float* dst) {
    const int B = m->cfg.batch_size;
    SERT_TRY(settle_entity_update(m));
    SERT_TRY(ensure_rw_current(m, -1));
    if (is_fs(m)) {
        SERT_TRY(fs_forward<false>(m, ds, batch_index));
    } else if (is_vs(m)) {
        SERT_TRY(vs_negatives(m, negatives, (uint64_t)(m->eval_draws++) * 2 + 1, m->stream));
        SERT_TRY(vs_project(m, ds, batch_index));
        SERT_TRY(vs_loss<false>(m, ds, batch_index));
    } else {
        SERT_TRY(ll_forward<false>(m, ds, batch_index));
    }
    const int nb = std::min(kOptBlocks, cdiv(B, 256));
    hipLaunchKernelGGL(sum_partial, dim3(nb), dim3(256), 0, m->stream, m->rowloss, (size_t)B, m->red_loss);
    hipLaunchKernelGGL(finalize_loss, dim3(1), dim3(256), 0, m->stream, m->red_loss, nb, m->red_loss, 0,
                       1.0f / (float)B, 0.0f, dst);
    return 0;
}

static int eval_check_args(sert_model* m, int split) {
    if (!m) SERT_FAIL("null model");
    if (split != SERT_SPLIT_TRAIN && split != SERT_SPLIT_VALIDATE) SERT_FAIL("bad split");
    if (m->cfg.inference_only) SERT_FAIL("model was created inference_only");
    return 0;
}

int sert_eval_batch(sert_model* m, int split, int64_t batch_index, const int64_t* negatives, float* loss_out) {
    refresh_gemm_choice();
    SERT_TRY(eval_check_args(m, split));
    invalidate_speculation(m);   // evaluation reuses the activation buffers and the negatives
    SERT_HIP(hipSetDevice(m->cfg.device));
    SERT_TRY(ensure_full_rw(m));
    const DataSplit& ds = m->split[split];
    const int B = m->cfg.batch_size;
    if (batch_index < 0 || (batch_index + 1) * (int64_t)B > ds.N) SERT_FAIL("batch_index out of range");
    SERT_TRY(eval_step_async(m, ds, batch_index, negatives, m->d_loss));
    SERT_HIP(hipMemcpyAsync(m->h_loss, m->d_loss, 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    float v = m->h_loss[0];
    if (is_dp(m)) {
        // eval loss of the global batch = mean of the per-rank means (equal shares)
        float* tmp = m->d_loss + 3;
        SERT_HIP(hipMemcpyAsync(tmp, m->d_loss, sizeof(float), hipMemcpyDeviceToDevice, m->stream));
        if (m->host_ar) SERT_TRY(host_allreduce(m, tmp, 1, m->stream));
        else            SERT_NCCL(g_rccl.AllReduce(tmp, tmp, 1, 7, 0, m->comm, m->stream));
        SERT_HIP(hipMemcpyAsync(m->h_loss, tmp, sizeof(float), hipMemcpyDeviceToHost, m->stream));
        SERT_HIP(hipStreamSynchronize(m->stream));
        v = m->h_loss[0] / (float)m->world;
    }
    timing_collect(m);
    if (loss_out) *loss_out = v;
    return 0;
}

int sert_eval_batches(sert_model* m, int split, const int64_t* batch_indices, int64_t count, float* losses_out) {
    refresh_gemm_choice();
    SERT_TRY(eval_check_args(m, split));
    if (!batch_indices || count < 0) SERT_FAIL("bad argument");
    if (count == 0) return 0;
    invalidate_speculation(m);
    SERT_HIP(hipSetDevice(m->cfg.device));
    SERT_TRY(ensure_full_rw(m));
    const DataSplit& ds = m->split[split];
    const int B = m->cfg.batch_size;
    for (int64_t i = 0; i < count; ++i)
        if (batch_indices[i] < 0 || (batch_indices[i] + 1) * (int64_t)B > ds.N) SERT_FAIL("batch_index out of range");
    if (m->d_losses_cap < count) {
        (void)hipFree(m->d_losses); m->d_losses = nullptr; m->d_losses_cap = 0;
        SERT_TRY(dmalloc(&m->d_losses, (size_t)count * 3));
        m->d_losses_cap = count;
    }
    // every batch writes its mean to its own slot: the host synchronises once per call
    for (int64_t i = 0; i < count; ++i) {
        SERT_TRY(eval_step_async(m, ds, batch_indices[i], nullptr, m->d_losses + 3 * i));
        if (m->timing.enabled) {  // events are single-slot: drain per step when timing
            SERT_HIP(hipStreamSynchronize(m->stream));
            timing_collect(m);
        }
    }
    if (is_dp(m)) {
        // global mean = mean of the per-rank means (equal shares): one collective over all slots
        if (m->host_ar) SERT_TRY(host_allreduce(m, m->d_losses, (size_t)count * 3, m->stream));
        else            SERT_NCCL(g_rccl.AllReduce(m->d_losses, m->d_losses, (size_t)count * 3, 7, 0, m->comm, m->stream));
    }
    std::vector<float> tmp((size_t)count * 3);
    SERT_HIP(hipMemcpyAsync(tmp.data(), m->d_losses, count * 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    const float scale = is_dp(m) ? 1.0f / (float)m->world : 1.0f;
    if (losses_out)
        for (int64_t i = 0; i < count; ++i) losses_out[i] = tmp[3 * i] * scale;
    return 0;
}

// grow-only device scratch (floats)
static int pred_reserve(float** buf, size_t* cap, size_t count) {
    if (*cap >= count) return 0;
    (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    SERT_TRY(dmalloc(buf, count));
    *cap = count;
    return 0;
}

int sert_predict_project(sert_model* m, const float* avg, int64_t Q, float* out) {
    refresh_gemm_choice();
    if (!m || !avg || !out) SERT_FAIL("null argument");
    if (!is_vs(m)) SERT_FAIL("sert_predict_project is the vectorspace predict_fn");
    if (Q <= 0) return 0;
    SERT_HIP(hipSetDevice(m->cfg.device));
    const int dw = m->cfg.word_dim, de = m->cfg.entity_dim;
    SERT_HIP(hipStreamSynchronize(m->stream));          // (the scratch may be in use by an earlier call's copy)
    SERT_TRY(pred_reserve(&m->pred_a, &m->pred_a_cap, (size_t)Q * dw));
    SERT_TRY(pred_reserve(&m->pred_b, &m->pred_b_cap, (size_t)Q * de));
    SERT_HIP(hipMemcpyAsync(m->pred_a, avg, (size_t)Q * dw * sizeof(float), hipMemcpyHostToDevice, m->stream));
    launch_gemm<false, false, EPI_BIAS_TANH>(m->stream, m->pred_a, m->W, m->pred_b, m->b, (int)Q, de, dw, dw, de, de);
    SERT_HIP(hipMemcpyAsync(out, m->pred_b, (size_t)Q * de * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    return 0;
}

int sert_predict_tokens(sert_model* m, const void* ids, int64_t rows, float* out) {
    refresh_gemm_choice();
    if (!m || !ids || !out) SERT_FAIL("null argument");
    if (is_vs(m)) SERT_FAIL("sert_predict_tokens is the loglinear predict_fn");
    if (rows <= 0) return 0;
    SERT_HIP(hipSetDevice(m->cfg.device));
    SERT_TRY(ensure_full_rw(m));
    SERT_TRY(ensure_rw_current(m, -1));
    const auto& c = m->cfg;
    const int n = c.window_size, d = c.word_dim, V = c.num_entities;
    const int64_t toks = rows * n;
    {
        bool ok = true;
        SERT_ID_DISPATCH(c.id_bytes, {
            const IdT* xi = (const IdT*)ids;
            for (int64_t i = 0; i < toks && ok; ++i) ok = (uint32_t)xi[i] < (uint32_t)c.vocab_size;
        });
        if (!ok) SERT_FAIL("token id >= vocab_size in ids");
    }
    SERT_HIP(hipStreamSynchronize(m->stream));
    if (m->pred_ids_cap < (size_t)toks * c.id_bytes) {
        (void)hipFree(m->pred_ids);
        m->pred_ids = nullptr; m->pred_ids_cap = 0;
        SERT_HIP(hipMalloc(&m->pred_ids, (size_t)toks * c.id_bytes));
        m->pred_ids_cap = (size_t)toks * c.id_bytes;
    }
    SERT_TRY(pred_reserve(&m->pred_a, &m->pred_a_cap, (size_t)toks * d));
    SERT_TRY(pred_reserve(&m->pred_b, &m->pred_b_cap, (size_t)toks * V));
    float *dG = m->pred_a, *dZ = m->pred_b;
    SERT_HIP(hipMemcpyAsync(m->pred_ids, ids, (size_t)toks * c.id_bytes, hipMemcpyHostToDevice, m->stream));
    SERT_ID_DISPATCH(c.id_bytes, {
        const IdT* X = (const IdT*)m->pred_ids;
        if (d % 4 == 0)
            hipLaunchKernelGGL((ll_gather_rows<IdT, 4>), dim3(grid_for(toks * d / 4, 256, 1 << 20)), dim3(256), 0, m->stream, X, m->rw, dG, toks, d);
        else
            hipLaunchKernelGGL((ll_gather_rows<IdT, 1>), dim3(grid_for(toks * d, 256, 1 << 20)), dim3(256), 0, m->stream, X, m->rw, dG, toks, d);
    });
    launch_gemm<false, false, EPI_BIAS>(m->stream, dG, m->W, dZ, m->b, (int)toks, V, d, d, V, V);
    hipLaunchKernelGGL(ll_softmax_rows, dim3(cdiv(toks, 4)), dim3(256), 0, m->stream, dZ, toks, V);
    SERT_HIP(hipMemcpyAsync(out, dZ, (size_t)toks * V * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream));
    return 0;
}

int sert_scorer_create(int device, const float* entities, int64_t V, int32_t dim, sert_scorer** out) {
    if (!entities || !out) SERT_FAIL("null argument");
    if (V <= 0 || dim <= 0) SERT_FAIL("bad sizes");
    SERT_HIP(hipSetDevice(device));
    sert_scorer* sc = new sert_scorer();
    sc->device = device;
    sc->V = V;
    sc->dim = dim;
    SERT_HIP(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
    SERT_HIP(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
    SERT_HIP(hipEventCreateWithFlags(&sc->ev_ready, hipEventDisableTiming));
    SERT_HIP(hipEventCreateWithFlags(&sc->ev_done, hipEventDisableTiming));
    SERT_TRY(dmalloc(&sc->E, (size_t)V * dim));
    SERT_HIP(hipMemcpyAsync(sc->E, entities, (size_t)V * dim * sizeof(float), hipMemcpyHostToDevice, sc->stream));
    hipLaunchKernelGGL(l2_normalize_rows, dim3(cdiv(V, 4)), dim3(256), 0, sc->stream, sc->E, V, dim);
    // large tables: bf16 copy for the prefilter GEMM (SERT_SCORE_FP32=1 keeps the fp32 filter)
    static const bool fp32_only = knob("SERT_SCORE_FP32") != nullptr;
    sc->bf16 = !fp32_only && V >= 32768 && dim % 4 == 0;
    if (sc->bf16) {
        sc->kp = (int)round_up(dim, 32);
        SERT_TRY(dmalloc(&sc->E16, (size_t)V * sc->kp));
        hipLaunchKernelGGL(to_bf16_rows, dim3(grid_for(V * sc->kp)), dim3(256), 0, sc->stream, sc->E, V, dim,
                           sc->kp, sc->E16);
    }
    SERT_HIP(hipStreamSynchronize(sc->stream));
    *out = sc;
    return 0;
}

int sert_scorer_destroy(sert_scorer* sc) {
    if (!sc) return 0;
    (void)hipSetDevice(sc->device);
    (void)hipFree(sc->E); (void)hipFree(sc->P); (void)hipFree(sc->S); (void)hipFree(sc->val); (void)hipFree(sc->idx);
    (void)hipFree(sc->Ss); (void)hipFree(sc->thr); (void)hipFree(sc->cand); (void)hipFree(sc->cnt);
    (void)hipFree(sc->nflag); (void)hipFree(sc->flag_list); (void)hipFree(sc->Pc); (void)hipFree(sc->idx_c);
    (void)hipFree(sc->val_c); (void)hipFree(sc->E16); (void)hipFree(sc->P16);
    if (sc->ev_ready) (void)hipEventDestroy(sc->ev_ready);
    if (sc->ev_done) (void)hipEventDestroy(sc->ev_done);
    if (sc->stream2) (void)hipStreamDestroy(sc->stream2);
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    delete sc;
    return 0;
}

// SERT_SCORE_BIG_TILE=1 (opt-in, measured slower at d_e = 128 -- DESIGN.md): score through
// gemm_big.h (256x256 tiles, one wave per SIMD).  Both the fused path and its exact fallback
// then use that kernel, so a row's scores never depend on which path produced them.
static int scorer_big_tile(const sert_scorer* sc) {   // 0 no, 1 = 256x256, 2 = 256x128 (two workgroups per CU)
#ifdef SERT_VARIANTS
    static const int big_tile = variant_knob("SERT_SCORE_BIG_TILE") ? atoi(variant_knob("SERT_SCORE_BIG_TILE")) : 0;
    return (sc->V >= 32768 && gemm_big_ok(sc->dim, sc->dim, sc->dim)) ? big_tile : 0;
#else
    (void)sc;
    return 0;
#endif
}

// Materialising path: (QT, V) cosine slabs + per-row selection, for a device-resident
// block of normalised projections.  Query tiles alternate between two streams; on return
// everything is ordered on sc->stream.
static int scorer_topk_materialised(sert_scorer* sc, const float* P, int64_t Q, int k, int32_t* idx,
                                    float* val) {
    hipStream_t s = sc->stream;
    const int64_t V = sc->V;
    const int dim = sc->dim;
    static const int64_t slab_elems = [] {
        const char* e = variant_knob("SERT_SCORE_SLAB_MB");   // tuning knob
        const int64_t mb = e ? atoll(e) : 0;
        return mb > 0 ? (mb << 20) / 4 : ((int64_t)1 << 27);
    }();
    // query tile: bounds one materialised score slab to ~0.5 GiB (two slabs alternate)
    const int64_t QT = std::min<int64_t>(Q, std::max<int64_t>(128, slab_elems / V / 128 * 128));
    if (sc->cap_s < 2 * QT * V) {
        SERT_HIP(hipStreamSynchronize(s));
        (void)hipFree(sc->S);
        sc->S = nullptr; sc->cap_s = 0;
        SERT_TRY(dmalloc(&sc->S, (size_t)2 * QT * V));
        sc->cap_s = 2 * QT * V;
    }
    SERT_HIP(hipEventRecord(sc->ev_ready, s));
    SERT_HIP(hipStreamWaitEvent(sc->stream2, sc->ev_ready, 0));
    int t = 0;
    for (int64_t q0 = 0; q0 < Q; q0 += QT, ++t) {
        const int64_t qn = std::min(QT, Q - q0);
        hipStream_t st = (t & 1) ? sc->stream2 : s;
        float* S = sc->S + (size_t)(t & 1) * QT * V;
        // S = P.E^T  (cosines), then per-row selection
        // (same kernel family as the fused path, so a row's scores do not depend on the path)
#ifdef SERT_VARIANTS
        if (scorer_big_tile(sc))
            launch_gemm_big_nt(st, P + q0 * dim, sc->E, S, (int)qn, (int)V, dim, dim, dim, (int)V, scorer_big_tile(sc) == 2);
        else
#endif
            launch_gemm<false, true, EPI_STORE>(st, P + q0 * dim, sc->E, S, nullptr, (int)qn, (int)V, dim,
                                                dim, dim, (int)V);
        hipLaunchKernelGGL(topk_rows, dim3((unsigned)qn), dim3(256), 0, st, S, (int)V, k, idx + q0 * k,
                           val + q0 * k, (float*)nullptr);
        if (sc->bf16) {   // same exact_dot scores and order as the bf16-prefiltered path reports
            int sn = 2;
            while (sn < k) sn <<= 1;
            hipLaunchKernelGGL(rescore_topk_rows, dim3((unsigned)qn), dim3(256), (size_t)sn * sizeof(unsigned long long),
                               st, P + q0 * dim, sc->E, dim, k, idx + q0 * k, val + q0 * k);
        }
    }
    SERT_HIP(hipEventRecord(sc->ev_done, sc->stream2));
    SERT_HIP(hipStreamWaitEvent(s, sc->ev_done, 0));
    return 0;
}

// Fused path (kernels_score.h): sampled thresholds, GEMM with a filtering epilogue,
// selection from the candidate lists; flagged rows are redone by the materialising path.
// proj / idx_out / score_out: the caller's host arrays.  The projections are uploaded chunk
// by chunk and each chunk's results are copied out while later chunks compute; *copied_out
// tells the caller that the host arrays are complete (no row needed the exact fallback).
static int scorer_topk_fused(sert_scorer* sc, const float* proj, int64_t Q, int k, int rs, int32_t* idx_out,
                             float* score_out, bool* copied_out) {
    *copied_out = false;
    hipStream_t s = sc->stream;
    const int64_t V = sc->V;
    const int dim = sc->dim;
    const int64_t Vs = cdiv(V, kScoreStride);
    // Query chunks of <= 8192 rows alternate between two streams, each with its own set of
    // scratch buffers: the selection kernel of one chunk (latency / random-row bound) runs under
    // the filter GEMM of the next (VALU / L2 bound).  An even number of equal chunks.
    static const int64_t chunk_rows = variant_knob("SERT_SCORE_CHUNK") ? atoll(variant_knob("SERT_SCORE_CHUNK")) : 8192;   // tuning knob
    const int64_t nchunks = Q <= 1024 ? 1 : 2 * cdiv(Q, 2 * chunk_rows);
    const int64_t QT = std::min<int64_t>(Q, round_up(cdiv(Q, nchunks), 128));
    if (sc->cap_ss < 2 * QT * Vs) {
        (void)hipFree(sc->Ss); sc->Ss = nullptr; sc->cap_ss = 0;
        SERT_TRY(dmalloc(&sc->Ss, (size_t)(2 * QT * Vs)));
        sc->cap_ss = 2 * QT * Vs;
    }
    // per-(row, 64-entity group) candidate lists: 8 slots for ~0.5 expected entries per
    // group (k <= 128), 16 beyond
    const int ngroups = 2 * cdiv((int)V, GN);
    const int gcap = k <= 128 ? 8 : 16;
    if (sc->cap_ft < 2 * QT) {
        (void)hipFree(sc->thr); sc->thr = nullptr; sc->cap_ft = 0;
        SERT_TRY(dmalloc(&sc->thr, (size_t)(2 * QT)));
        sc->cap_ft = 2 * QT;
    }
    const int64_t cand_set = QT * ngroups * gcap, cnt_set = QT * ngroups;
    if (sc->cap_cand < 2 * cand_set) {
        (void)hipFree(sc->cand); (void)hipFree(sc->cnt);
        sc->cand = nullptr; sc->cnt = nullptr; sc->cap_cand = 0;
        SERT_TRY(dmalloc(&sc->cand, (size_t)(2 * cand_set)));
        SERT_TRY(dmalloc(&sc->cnt, (size_t)(2 * QT * ngroups * 16 / 8)));   // sized for either gcap
        sc->cap_cand = 2 * cand_set;
    }
    if (sc->cap_flag < Q) {
        (void)hipFree(sc->flag_list); (void)hipFree(sc->nflag);
        sc->flag_list = nullptr; sc->nflag = nullptr; sc->cap_flag = 0;
        SERT_TRY(dmalloc(&sc->flag_list, (size_t)Q));
        SERT_TRY(dmalloc(&sc->nflag, (size_t)1));
        sc->cap_flag = Q;
    }
    // the bf16 prefilter needs a gap of 2 delta between the k-th score and the filter threshold;
    // a table whose rows mostly lack it (very high d_e, heavy ties) is scored in fp32 from then on
    const bool use_bf16 = sc->bf16 && !sc->bf16_demoted;
    if (use_bf16 && sc->cap_p16 < 2 * QT * sc->kp) {
        (void)hipFree(sc->P16); sc->P16 = nullptr; sc->cap_p16 = 0;
        SERT_TRY(dmalloc(&sc->P16, (size_t)(2 * QT * sc->kp)));
        sc->cap_p16 = 2 * QT * sc->kp;
    }
    SERT_HIP(hipMemsetAsync(sc->nflag, 0, sizeof(int), s));
    SERT_HIP(hipEventRecord(sc->ev_ready, s));           // nflag zeroed, earlier work on s done
    SERT_HIP(hipStreamWaitEvent(sc->stream2, sc->ev_ready, 0));
    auto copy_out = [&](int64_t q0, int64_t qn, hipStream_t st) {   // (pageable destination: returns when done)
        hipError_t e = hipMemcpyAsync(idx_out + q0 * k, sc->idx + q0 * k, (size_t)qn * k * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(score_out + q0 * k, sc->val + q0 * k, (size_t)qn * k * sizeof(float), hipMemcpyDeviceToHost, st);
        return e;
    };
    int64_t t = 0;
    for (int64_t q0 = 0; q0 < Q; q0 += QT, ++t) {
        const int64_t qn = std::min(QT, Q - q0);
        const int set = (int)(t & 1);
        hipStream_t st = set ? sc->stream2 : s;
        float* Pw = sc->P + q0 * dim;
        SERT_HIP(hipMemcpyAsync(Pw, proj + q0 * dim, (size_t)qn * dim * sizeof(float), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(l2_normalize_rows, dim3(cdiv(qn, 4)), dim3(256), 0, st, Pw, qn, dim);
        const float* P = Pw;
        float* Ss = sc->Ss + (size_t)set * QT * Vs;
        float* thr = sc->thr + (size_t)set * QT;
        unsigned long long* cand = sc->cand + (size_t)set * cand_set;
        unsigned char* cnt = sc->cnt + (size_t)set * cnt_set;
        uint16_t* P16 = use_bf16 ? sc->P16 + (size_t)set * QT * sc->kp : nullptr;
        SERT_HIP(hipMemsetAsync(cnt, 0, (size_t)qn * ngroups, st));
        // 1. cosines against every kScoreStride-th entity; threshold = rs-th best of the sample
        // (bf16 scorer: approximate sample scores are as good for choosing a threshold)
        if (use_bf16) {
            hipLaunchKernelGGL(to_bf16_rows, dim3(grid_for(qn * sc->kp)), dim3(256), 0, st, P, qn, dim, sc->kp, P16);
            launch_score_sample_bf16(st, P16, sc->E16, Ss, (int)qn, (int)Vs, sc->kp, kScoreStride);
        } else
            launch_gemm<false, true, EPI_STORE>(st, P, sc->E, Ss, nullptr, (int)qn, (int)Vs, dim, dim,
                                                dim * kScoreStride, (int)Vs);
        if (rs <= 64 && Vs >= 2048)
            hipLaunchKernelGGL(approx_kth_rows, dim3((unsigned)qn), dim3(256), 0, st, Ss, (int)Vs, rs, thr);
        else
            hipLaunchKernelGGL(kth_largest_rows, dim3((unsigned)qn), dim3(256), 0, st, Ss, (int)Vs, rs, thr);
        // 2. full GEMM, filtering epilogue
        if (use_bf16) {
            launch_score_filter_bf16(st, P16, sc->E16, thr, (uint32_t*)cand, cnt, ngroups, gcap, (int)qn, (int)V, sc->kp);
        } else
#ifdef SERT_VARIANTS
        if (scorer_big_tile(sc))
            launch_gemm_big_filter(st, P, sc->E, thr, cand, cnt, ngroups, gcap, (int)qn, (int)V, dim, dim, dim,
                                   scorer_big_tile(sc) == 2);
        else
#endif
            launch_gemm<false, true, EPI_FILTER>(st, P, sc->E, nullptr, thr, (int)qn, (int)V, dim, dim, dim,
                                                 (int)V, 1, 0, 0, cand, cnt, gcap);
        // 3. selection from the candidate lists
        // candidate capacity: expected 2k+400, sigma ~ 16 sqrt(rs): the next power of two above +6 sigma
        int ccap = 1024;
        while (ccap < 2 * k + 400 + 6 * 16 * (int)ceilf(sqrtf((float)rs)) && ccap < kCandCap) ccap <<= 1;
        if (use_bf16)
            hipLaunchKernelGGL(topk_from_groups_rescore, dim3((unsigned)qn), dim3(256), (size_t)ccap * sizeof(unsigned long long), st,
                               (const uint32_t*)cand, cnt, ngroups, gcap, k, sc->idx + q0 * k, sc->val + q0 * k, (int)q0,
                               sc->nflag, sc->flag_list, ccap, P, sc->E, dim, thr, bf16_delta(dim));
        else
            hipLaunchKernelGGL(topk_from_groups, dim3((unsigned)qn), dim3(256), (size_t)ccap * sizeof(unsigned long long), st,
                               cand, cnt, ngroups, gcap, k, sc->idx + q0 * k, sc->val + q0 * k, (int)q0,
                               sc->nflag, sc->flag_list, ccap);
        // results of the previous chunk (other stream) travel while this one computes
        if (t >= 1) SERT_HIP(copy_out(q0 - QT, QT, set ? s : sc->stream2));
    }
    {
        const int64_t q_last = (t - 1) * QT;
        SERT_HIP(copy_out(q_last, Q - q_last, ((t - 1) & 1) ? sc->stream2 : s));
    }
    SERT_HIP(hipEventRecord(sc->ev_done, sc->stream2));
    SERT_HIP(hipStreamWaitEvent(s, sc->ev_done, 0));
    int nf = 0;
    SERT_HIP(hipMemcpyAsync(&nf, sc->nflag, sizeof(int), hipMemcpyDeviceToHost, s));
    SERT_HIP(hipStreamSynchronize(s));
    if (nf == 0) { *copied_out = true; return 0; }
    if (use_bf16 && (int64_t)nf * 4 > Q && Q >= 64) sc->bf16_demoted = true;
    // rows the sample misjudged: recompute exactly (ascending order, for reproducibility)
    std::vector<int> list((size_t)nf);
    SERT_HIP(hipMemcpy(list.data(), sc->flag_list, (size_t)nf * sizeof(int), hipMemcpyDeviceToHost));
    std::sort(list.begin(), list.end());
    SERT_HIP(hipMemcpyAsync(sc->flag_list, list.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice, s));
    if (sc->cap_c < nf) {
        (void)hipFree(sc->Pc); sc->Pc = nullptr; sc->cap_c = 0;
        SERT_TRY(dmalloc(&sc->Pc, (size_t)nf * dim));
        sc->cap_c = nf;
    }
    if (sc->cap_ck < (int64_t)nf * k) {
        (void)hipFree(sc->idx_c); (void)hipFree(sc->val_c);
        sc->idx_c = nullptr; sc->val_c = nullptr; sc->cap_ck = 0;
        SERT_TRY(dmalloc(&sc->idx_c, (size_t)nf * k));
        SERT_TRY(dmalloc(&sc->val_c, (size_t)nf * k));
        sc->cap_ck = (int64_t)nf * k;
    }
    hipLaunchKernelGGL(gather_rows_f32, dim3(grid_for((int64_t)nf * dim)), dim3(256), 0, s, sc->P,
                       sc->flag_list, nf, dim, sc->Pc);
    SERT_TRY(scorer_topk_materialised(sc, sc->Pc, nf, k, sc->idx_c, sc->val_c));
    hipLaunchKernelGGL(scatter_topk_rows, dim3(grid_for((int64_t)nf * k)), dim3(256), 0, s, sc->idx_c,
                       sc->val_c, sc->flag_list, nf, k, sc->idx, sc->val);
    return 0;
}

int sert_scorer_topk(sert_scorer* sc, const float* proj, int64_t Q, int32_t k, int32_t* idx_out, float* score_out) {
    if (!sc || !proj || !idx_out || !score_out) SERT_FAIL("null argument");
    if (Q < 0 || k <= 0) SERT_FAIL("bad sizes");
    if (k > sc->V) SERT_FAIL("k exceeds the number of entities");
    if (k > kTopKMax) SERT_FAIL("k > 1024 is not supported");
    if (Q == 0) return 0;
    SERT_HIP(hipSetDevice(sc->device));
    hipStream_t s = sc->stream;
    const int64_t V = sc->V;
    const int dim = sc->dim;
    if (sc->cap_q < Q) {
        (void)hipFree(sc->P); (void)hipFree(sc->val); (void)hipFree(sc->idx);
        sc->P = nullptr; sc->val = nullptr; sc->idx = nullptr;
        sc->cap_q = 0; sc->cap_qk = 0;
        SERT_TRY(dmalloc(&sc->P, (size_t)Q * dim));
        sc->cap_q = Q;
    }
    if (sc->cap_qk < Q * k) {
        (void)hipFree(sc->val); (void)hipFree(sc->idx);
        sc->val = nullptr; sc->idx = nullptr; sc->cap_qk = 0;
        SERT_TRY(dmalloc(&sc->val, (size_t)Q * k));
        SERT_TRY(dmalloc(&sc->idx, (size_t)Q * k));
        sc->cap_qk = Q * k;
    }
    // fused path for large entity tables: the sample must be big enough for a stable
    // threshold: rank rs among the V/16 sampled entities, i.e. an expected 2k+400 (std ~
    // sqrt(rs)*16) candidates of V -- at k=100: 608 +- 99, >= k at 5 sigma, <= 1024 at 4
    const int rs = cdiv(2 * k + 400, kScoreStride);
    static const bool never_fuse = knob("SERT_SCORE_MATERIALISE") != nullptr;   // cross-check knob
    const bool fused = !never_fuse && V >= 32768 && dim % 4 == 0 && rs <= kTopKMax &&
                       cdiv(V, kScoreStride) >= 8 * (int64_t)rs;
    bool copied = false;
    if (fused) SERT_TRY(scorer_topk_fused(sc, proj, Q, k, rs, idx_out, score_out, &copied));
    else {
        SERT_HIP(hipMemcpyAsync(sc->P, proj, (size_t)Q * dim * sizeof(float), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(l2_normalize_rows, dim3(cdiv(Q, 4)), dim3(256), 0, s, sc->P, Q, dim);
        SERT_TRY(scorer_topk_materialised(sc, sc->P, Q, k, sc->idx, sc->val));
    }
    SERT_HIP(hipGetLastError());
    if (copied) return 0;
    SERT_HIP(hipMemcpyAsync(idx_out, sc->idx, (size_t)Q * k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SERT_HIP(hipMemcpyAsync(score_out, sc->val, (size_t)Q * k * sizeof(float), hipMemcpyDeviceToHost, s));
    SERT_HIP(hipStreamSynchronize(s));
    return 0;
}

int sert_scorer_scores(sert_scorer* sc, const float* proj, int64_t Q, float* score_out) {
    if (!sc || !proj || !score_out) SERT_FAIL("null argument");
    if (Q <= 0) return 0;
    SERT_HIP(hipSetDevice(sc->device));
    hipStream_t s = sc->stream;
    const int64_t V = sc->V;
    const int dim = sc->dim;
    const int64_t QT = std::min<int64_t>(Q, std::max<int64_t>(128, ((int64_t)1 << 28) / V / 128 * 128));
    if (sc->cap_q < Q) {
        (void)hipFree(sc->P); (void)hipFree(sc->val); (void)hipFree(sc->idx);
        SERT_TRY(dmalloc(&sc->P, (size_t)Q * dim));
        sc->cap_q = Q; sc->cap_qk = 0; sc->val = nullptr; sc->idx = nullptr;
    }
    if (sc->cap_s < QT * V) {
        (void)hipFree(sc->S);
        SERT_TRY(dmalloc(&sc->S, (size_t)QT * V));
        sc->cap_s = QT * V;
    }
    SERT_HIP(hipMemcpyAsync(sc->P, proj, (size_t)Q * dim * sizeof(float), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(l2_normalize_rows, dim3(cdiv(Q, 4)), dim3(256), 0, s, sc->P, Q, dim);
    for (int64_t q0 = 0; q0 < Q; q0 += QT) {
        const int64_t qn = std::min(QT, Q - q0);
        launch_gemm<false, true, EPI_STORE>(s, sc->P + q0 * dim, sc->E, sc->S, nullptr, (int)qn, (int)V, dim,
                                            dim, dim, (int)V);
        hipLaunchKernelGGL(cos_to_score, dim3(grid_for(qn * V)), dim3(256), 0, s, sc->S, (size_t)(qn * V));
        SERT_HIP(hipMemcpyAsync(score_out + q0 * V, sc->S, (size_t)qn * V * sizeof(float), hipMemcpyDeviceToHost, s));
        SERT_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

int sert_host_alloc(void** out, size_t bytes) {
    if (!out || bytes == 0) SERT_FAIL("bad argument");
    *out = nullptr;
    static const int flags = variant_knob("SERT_PIN_FLAGS") ? atoi(variant_knob("SERT_PIN_FLAGS")) : (int)hipHostMallocDefault;   // tuning knob
    SERT_HIP(hipHostMalloc(out, bytes, (unsigned)flags));
    return 0;
}
int sert_host_free(void* p) {
    if (p) SERT_HIP(hipHostFree(p));
    return 0;
}

int sert_score_topk(int device, const float* entities, int64_t V, int32_t dim, const float* proj,
                    int64_t Q, int32_t k, int32_t* idx_out, float* score_out) {
    sert_scorer* sc = nullptr;
    SERT_TRY(sert_scorer_create(device, entities, V, dim, &sc));
    const int rc = sert_scorer_topk(sc, proj, Q, k, idx_out, score_out);
    sert_scorer_destroy(sc);
    return rc;
}

// Slabs per big tensor (model.h).  One by default: every extra collective adds its own start-up
// latency to the exchange, which on a world of one costs more than the overlap of slab c's
// optimiser with slab c+1's reduce-scatter returns; SERT_AR_CHUNKS=k is there to be tuned on a
// multi-GPU node.
// The word table is exchanged by rows (kernels_xchg.h) unless SERT_DP_EXCHANGE=zero1 asks for the
// reduce-scatter / all-gather of whole slabs, or the model cannot take it: rows that are no multiple of
// 16 bytes, gradients the caller wants to read back (keep_grads), several slabs per tensor.
static bool row_exchange_wanted(const sert_model* m) {
    const char* e = knob("SERT_DP_EXCHANGE");
    if (e && !strcmp(e, "zero1")) return false;
    return m->cfg.word_dim % 4 == 0 && !m->cfg.keep_grads && m->ar_chunks == 1 && !m->cfg.inference_only;
}

static int exchange_slabs() {
    const char* e = knob("SERT_AR_CHUNKS");
    const int want = e ? atoi(e) : 1;
    return std::max(1, std::min(want, (int)sert_model::kMaxArChunks));
}

// (see create_intra_events: a model that gets a communicator takes system-scope events; nothing is in flight afterwards)
static int comm_scope_events(sert_model* m) {
    if (!m->events_device_scope) return 0;
    SERT_HIP(hipDeviceSynchronize());
    return create_intra_events(m, false);
}

int sert_comm_unique_id(char id[SERT_COMM_ID_BYTES]) {
    SERT_TRY(rccl_load());
    SERT_NCCL(g_rccl.GetUniqueId(id));
    return 0;
}

int sert_comm_init(sert_model* m, const char id[SERT_COMM_ID_BYTES], int rank, int world) {
    if (!m || !id) SERT_FAIL("null argument");
    if (world < 1 || rank < 0 || rank >= world) SERT_FAIL("bad rank/world");
    if ((int64_t)m->cfg.batch_size * world != m->cfg.global_batch_size)
        SERT_FAIL("global_batch_size must equal batch_size * world");
    SERT_TRY(rccl_load());
    SERT_HIP(hipSetDevice(m->cfg.device));
    invalidate_speculation(m);
    SERT_TRY(comm_scope_events(m));
    UniqueId uid;
    memcpy(uid.internal, id, SERT_COMM_ID_BYTES);
    for (int i = 0; i < 4; ++i)
        if (m->pt_sharded[i]) SERT_FAIL("this model already has a data-parallel communicator");
    SERT_NCCL(g_rccl.CommInitRank(&m->comm, world, uid, rank));
    m->rank = rank;
    m->world = world;
    if (!m->comm_stream) {
        SERT_HIP(hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking));
        SERT_HIP(hipEventCreateWithFlags(&m->ev_rest_ready, hipEventDisableTiming));
        SERT_HIP(hipEventCreateWithFlags(&m->ev_ar_done, hipEventDisableTiming));
        SERT_HIP(hipEventCreateWithFlags(&m->ev_ag_done, hipEventDisableTiming));
        for (int i = 0; i < 4; ++i) {
            SERT_HIP(hipEventCreateWithFlags(&m->ev_grad_ready[i], hipEventDisableTiming));
            for (int c = 0; c < sert_model::kMaxArChunks; ++c) {
                SERT_HIP(hipEventCreateWithFlags(&m->ev_rs_done[i][c], hipEventDisableTiming));
                SERT_HIP(hipEventCreateWithFlags(&m->ev_opt_done[i][c], hipEventDisableTiming));
            }
        }
    }
    m->ar_chunks = exchange_slabs();
    m->xr_mode = row_exchange_wanted(m);
    if (!m->ev_params_ready) {
        SERT_HIP(hipEventCreateWithFlags(&m->ev_params_ready, hipEventDisableTiming));
        SERT_HIP(hipEventCreateWithFlags(&m->ev_word_updated, hipEventDisableTiming));
    }
    const int rc = shard_setup(m);
    if (rc != 0) {   // never leave a communicator behind a model that could not be sharded
        const std::string why = g_last_error;
        (void)sert_comm_destroy(m);
        g_last_error = why;
    }
    return rc;
}

int sert_comm_init_host(sert_model* m, int rank, int world, sert_alltoall_fn fn, void* user) {
    if (!m || !fn) SERT_FAIL("null argument");
    if (world < 1 || rank < 0 || rank >= world) SERT_FAIL("bad rank/world");
    if ((int64_t)m->cfg.batch_size * world != m->cfg.global_batch_size)
        SERT_FAIL("global_batch_size must equal batch_size * world");
    if (m->comm) SERT_FAIL("an RCCL communicator is already attached");
    for (int i = 0; i < 4; ++i)
        if (m->pt_sharded[i]) SERT_FAIL("this model already has a data-parallel communicator");
    m->host_ar = fn;
    m->host_ar_user = user;
    m->rank = rank;
    m->world = world;
    invalidate_speculation(m);
    SERT_TRY(comm_scope_events(m));
    m->ar_chunks = exchange_slabs();
    m->xr_mode = row_exchange_wanted(m);
    const int rc = shard_setup(m);
    if (rc != 0) {
        const std::string why = g_last_error;
        (void)sert_comm_destroy(m);
        g_last_error = why;
    }
    return rc;
}

// After this call the model keeps its parameters (identical on every rank) but can no longer
// train or hand out its optimiser state: that state is sharded over ranks that are gone.
int sert_comm_destroy(sert_model* m) {
    if (!m) return 0;
    (void)hipSetDevice(m->cfg.device);
    if (m->comm_stream) (void)hipStreamSynchronize(m->comm_stream);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (!m->comm_dead && is_dp(m) && !m->rw_full) (void)ensure_full_rw(m);   // (collective: every rank destroys)
    if (m->host_ar || m->comm) m->comm_dead = true;
    m->host_ar = nullptr;
    m->host_ar_user = nullptr;
    if (m->comm) {
        void* cm = m->comm;
        m->comm = nullptr;
        SERT_NCCL(g_rccl.CommDestroy(cm));
    }
    return 0;
}

int sert_comm_stats(sert_model* m, double* out, int n) {
    if (!m || !out || n < 1 || n > 8) SERT_FAIL("bad argument");
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    v[0] = (double)m->world;
    if (is_dp(m) || m->comm_dead) {
        v[1] = m->pt_sharded[0] ? (m->xr_mode ? 2.0 : 1.0) : 0.0;
        v[2] = m->comm_steps > 0 ? m->comm_bytes_moved / (double)m->comm_steps : 0.0;
        const double W = (double)m->world;
        v[3] = 2.0 * 2.0 * (W - 1.0) / W * 4.0 * (double)m->pt_pad[0];
        v[4] = m->host_ar ? 2.0 : 1.0;
        v[5] = (double)m->comm_steps;
        if (m->xr && !m->xr->batches.empty()) {
            double f = 0.0, sv = 0.0;
            for (const RowExchangeBatch& b : m->xr->batches) { f += b.fetch_total; sv += b.serve_total; }
            v[6] = f / (double)m->xr->batches.size();
            v[7] = sv / (double)m->xr->batches.size();
        }
    }
    for (int i = 0; i < n; ++i) out[i] = v[i];
    return 0;
}

int sert_debug_poison_scratch(sert_model* m) {
    if (!m) SERT_FAIL("null argument");
    if (m->spec_fb_batch >= 0) SERT_FAIL("a run-ahead step is in flight (sert_hint_next_batch): its gradients live in the scratch");
    SERT_HIP(hipSetDevice(m->cfg.device));
    SERT_HIP(hipDeviceSynchronize());
    if (!m->gflat) return 0;
    // gradients, loss / sum-of-squares slots: quiet NaNs; the per-entity sorted-run bounds behind them: the in-range but wrong
    // run [0, 1) (a fix-up that trusted a stale bound would add chunk 0's carry to an entity the reduce never met)
    std::vector<uint32_t> h(m->gflat_alloc, 0x7fc00000u);
    if (m->run_start) {
        const size_t V4 = round_up((size_t)m->cfg.num_entities, 4);
        for (size_t i = 0; i < V4; ++i) { h[m->gflat_count + i] = 0u; h[m->gflat_count + V4 + i] = 1u; }
    }
    SERT_HIP(hipMemcpy(m->gflat, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return 0;
}

int sert_debug_update_counts(sert_model* m, int64_t* out, int n) {
    if (!m || !out || n < 1 || n > 10) SERT_FAIL("bad argument");
    for (int i = 0; i < n; ++i) out[i] = m->upd_counts[i];
    return 0;
}

int sert_debug_row_lists(const uint32_t* allbits, int world, int rank, int64_t num_batches, int64_t bit_words,
                         int64_t rows_per_rank, int64_t vocab, int64_t batch, int32_t* serve_cnt, int32_t* fetch_cnt,
                         int32_t* serve_rows, int32_t* fetch_rows, int32_t* union_rows, int32_t* ptr, int32_t* ent,
                         int64_t capacity, int64_t* sizes) {
    if (!allbits || world < 1 || rank < 0 || rank >= world || batch < 0 || batch >= num_batches || !sizes)
        SERT_FAIL("bad argument");
    RowExchangeLists L;
    build_row_exchange(allbits, world, rank, num_batches, bit_words, rows_per_rank, vocab, L);
    const RowExchangeBatch& xb = L.batches[(size_t)batch];
    sizes[0] = xb.serve_total; sizes[1] = xb.fetch_total; sizes[2] = xb.nunion; sizes[3] = xb.nent; sizes[4] = L.max_xfer_rows;
    if (std::max<int64_t>(std::max(xb.serve_total, xb.fetch_total), std::max(xb.nunion + 1, xb.nent)) > capacity)
        SERT_FAIL("capacity too small");
    for (int q = 0; q < world; ++q) { serve_cnt[q] = xb.serve_cnt[(size_t)q]; fetch_cnt[q] = xb.fetch_cnt[(size_t)q]; }
    std::copy_n(L.serve_rows.begin() + xb.serve_off, xb.serve_total, serve_rows);
    std::copy_n(L.fetch_rows.begin() + xb.fetch_off, xb.fetch_total, fetch_rows);
    std::copy_n(L.union_rows.begin() + xb.union_off, xb.nunion, union_rows);
    std::copy_n(L.ptr.begin() + xb.ptr_off, xb.nunion + 1, ptr);
    std::copy_n(L.ent.begin() + xb.ent_off, xb.nent, ent);
    return 0;
}

// Host only (no GPU): the per-batch inverted index of sert_upload_dataset (word_index.h) built for `ids` and EVALUATED
// on the host exactly as the kernels of kernels_seg.h walk it -- every item sums its entries left to right, final items
// store acc / divisor into the gradient table, chunk items into their partial row, the upper levels read the partial
// rows of the level below -- so the structure (levels, chunk bounds, partial-row numbering, the row-grouped level 0 with
// its eight XCD lists, the dense heavy words left out of the tree) is checked by the CPU suite.
int sert_debug_word_index_sum(const void* ids, int id_bytes, int64_t num_batches, int B, int n, int vocab, int row_groups,
                              int dense_heavy, int64_t batch, const float* src, int d, float divisor, float* grad_out,
                              int64_t* stats) {
    if (!ids || !src || !grad_out || !stats || num_batches <= 0 || B <= 0 || n <= 0 || vocab <= 0 || d <= 0 || batch < 0 ||
        batch >= num_batches || (id_bytes != 1 && id_bytes != 2 && id_bytes != 4))
        SERT_FAIL("bad argument");
    WordIndex wi;
    bool ok = true;
    SERT_ID_DISPATCH(id_bytes, ok = build_word_index<IdT>((const IdT*)ids, num_batches, B, n, vocab, /*row_is_pos=*/false, wi,
                                                          /*want_slots=*/false, (dense_heavy & 1) != 0, row_groups,
                                                          /*sort_level0=*/(dense_heavy & 2) != 0));
    if (!ok) SERT_FAIL("token id >= vocab");
    const BatchIndex& bx = wi.batches[(size_t)batch];
    // (dense_heavy & 2: level 0 sorted by item length, first row number in the descriptor -- what sert_upload_dataset builds
    //  for the vectorspace models; a row-grouped index is never sorted)
    if (((dense_heavy & 2) != 0 && row_groups <= 1 && bx.nlevels >= 1 && bx.item_cnt[0] > 0) != bx.slot_is_row) SERT_FAIL("slot_is_row not as asked");
    if (bx.slot_is_row) {
        const SegItem* l0 = wi.items.data() + bx.item_off[0];
        const int32_t* r0 = wi.rows.data() + bx.rows_off;
        for (int32_t k = 0; k < bx.item_cnt[0]; ++k) {
            if (l0[k].slot != r0[l0[k].begin]) SERT_FAIL("a level-0 item's slot is not its first row");
            if (k > 0 && l0[k].end - l0[k].begin > l0[k - 1].end - l0[k - 1].begin) SERT_FAIL("level 0 is not sorted by length");
        }
    }
    std::fill(grad_out, grad_out + (size_t)vocab * d, 0.f);
    std::vector<float> part((size_t)std::max<int64_t>(1, bx.part_rows) * d, 0.f);
    int64_t items_total = 0, finals = 0;
    for (int l = 0; l < bx.nlevels; ++l) {
        const SegItem* items = wi.items.data() + bx.item_off[l];
        const float* in = l == 0 ? src : part.data() + (size_t)bx.part_off[l - 1] * d;
        const int32_t* rows = l == 0 ? wi.rows.data() + bx.rows_off : nullptr;
        float* pout = part.data() + (size_t)bx.part_off[l] * d;
        // level 0 of a row-grouped index is addressed through its XCD lists, as the kernel does
        std::vector<int32_t> order;
        if (l == 0 && bx.row_groups > 1) {
            for (int x = 0; x < 8; ++x)
                for (int k = 0; k < bx.xcd_cnt[x]; ++k) order.push_back(bx.xcd_off[x] + k);
            if ((int32_t)order.size() != bx.item_cnt[0]) SERT_FAIL("XCD lists do not cover level 0");
        } else {
            for (int32_t k = 0; k < bx.item_cnt[l]; ++k) order.push_back(k);
        }
        for (int32_t k : order) {
            const SegItem& it = items[k];
            if (it.end - it.begin > kSegChunk && l < kSegMaxLevels - 1) SERT_FAIL("an item longer than a chunk");
            ++items_total;
            float* dst = it.dst >= 0 ? grad_out + (size_t)it.dst * d : pout + (size_t)(-(it.dst + 1)) * d;
            if (it.dst >= 0) ++finals;
            for (int c = 0; c < d; ++c) {
                float a = 0.f;
                if (l == 0 && bx.slot_is_row && it.end - it.begin == 1) a += in[(size_t)it.slot * d + c];   // (as segsum_rows does)
                else
                for (int32_t e = it.begin; e < it.end; ++e) a += in[(size_t)(rows ? rows[e] : e) * d + c];
                dst[c] = it.dst >= 0 ? a / divisor : a;
            }
        }
    }
    // the dense heavy words: count-weighted sums over all batch rows (segsum_heavy + combine; the block structure of the
    // device reduction is not restated here -- integer-valued test data makes every association exact)
    for (int h = 0; h < bx.dense_cnt; ++h) {
        const uint8_t* dc = wi.dense_counts.data() + (size_t)(batch * B) * kHeavyMax;
        for (int c = 0; c < d; ++c) {
            float a = 0.f;
            for (int i = 0; i < B; ++i) a += (float)dc[(size_t)i * kHeavyMax + h] * src[(size_t)i * d + c];
            grad_out[(size_t)bx.dense_word[h] * d + c] = a / divisor;
        }
    }
    stats[0] = bx.nlevels; stats[1] = items_total; stats[2] = bx.part_rows; stats[3] = finals; stats[4] = bx.dense_cnt;
    stats[5] = bx.row_groups; stats[6] = bx.item_cnt[0]; stats[7] = bx.num_distinct;
    return 0;
}

int sert_profile_range_push(const char* name) {
    roctx_load();
    if (!name) SERT_FAIL("null range name");
    return g_roctx.push ? (g_roctx.push(name) < 0 ? 1 : 0) : 0;
}
int sert_profile_range_pop(void) {
    roctx_load();
    return g_roctx.pop ? (g_roctx.pop() < 0 ? 1 : 0) : 0;
}

int sert_synchronize(sert_model* m) {
    if (!m) SERT_FAIL("null model");
    SERT_HIP(hipSetDevice(m->cfg.device));
    SERT_HIP(hipStreamSynchronize(m->stream));
    SERT_HIP(hipStreamSynchronize(m->stream2));
    if (m->stream3) SERT_HIP(hipStreamSynchronize(m->stream3));
    if (m->stream4) SERT_HIP(hipStreamSynchronize(m->stream4));
    if (m->comm_stream) SERT_HIP(hipStreamSynchronize(m->comm_stream));
    return 0;
}

int sert_timing_enable(sert_model* m, int on) {
    if (!m) SERT_FAIL("null model");
    if (on < 0 || on > 2) SERT_FAIL("timing mode: 0 = off, 1 = every group alone on one queue, 2 = in the step");
    SERT_HIP(hipSetDevice(m->cfg.device));
    if (m->instep.on && on != 2) {        // leaving the in-step mode: everything in flight is measured first
        SERT_HIP(hipDeviceSynchronize());
        instep_harvest(m, true);
    }
    if (on == 2 && !m->instep.created) {
        for (int i = 0; i < InStep::kRing; ++i)
            for (int k = 0; k < 2; ++k) SERT_HIP(hipEventCreate(&m->instep.ev[i][k]));
        m->instep.created = true;
    }
    m->timing.enabled = on == 1;
    m->instep.on = on == 2;
    m->instep.cur_group = -1;
    return 0;
}
int sert_timing_reset(sert_model* m) {
    if (!m) SERT_FAIL("null model");
    for (int g = 0; g < TG_COUNT; ++g) { m->timing.total_us[g] = 0; m->timing.samples[g] = 0; m->timing.used[g] = false; }
    if (m->instep.created) {
        SERT_HIP(hipSetDevice(m->cfg.device));
        SERT_HIP(hipDeviceSynchronize());
        instep_harvest(m, true);
        for (int g = 0; g < TG_COUNT; ++g) { m->instep.total_us[g] = 0; m->instep.launches[g] = 0; }
        m->instep.steps = 0;
    }
    return 0;
}
int sert_timing_count(sert_model*) { return TG_COUNT; }
const char* sert_timing_name(sert_model*, int i) { return (i >= 0 && i < TG_COUNT) ? kTimingNames[i] : ""; }
double sert_timing_avg_us(sert_model* m, int i) {
    if (!m || i < 0 || i >= TG_COUNT) return 0.0;
    if (m->instep.steps > 0) {
        // in-step mode: the group's kernel time per training step (sum of its timed launches' own durations)
        if (m->instep.head < m->instep.tail) {
            (void)hipSetDevice(m->cfg.device);
            (void)hipDeviceSynchronize();
            instep_harvest(m, true);
        }
        return m->instep.total_us[i] / (double)m->instep.steps;
    }
    if (m->timing.samples[i] == 0) return 0.0;
    return m->timing.total_us[i] / (double)m->timing.samples[i];
}
double sert_timing_launches(sert_model* m, int i) {
    if (!m || i < 0 || i >= TG_COUNT || m->instep.steps == 0) return 0.0;
    return (double)m->instep.launches[i] / (double)m->instep.steps;
}

int sert_bench_gemm(int device, int ta, int tb, int epi, int M, int N, int K, int splits, int iters,
                    double* avg_us) {
    refresh_gemm_choice();
    if (!avg_us || M <= 0 || N <= 0 || K <= 0 || iters <= 0) SERT_FAIL("bad argument");
    SERT_HIP(hipSetDevice(device));
    hipStream_t s;
    SERT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (splits < 1) splits = 1;
    int kper = (int)round_up(cdiv(K, splits), GK);
    splits = cdiv(K, kper);
    // SERT_BENCH_GEMM_CSB=1 (A^T.B only): with the column sums of B riding along, as dW + db run in the step
    const bool csb = ta && !tb && variant_knob("SERT_BENCH_GEMM_CSB") != nullptr;
    const size_t na = (size_t)M * K, nb = (size_t)K * N, nc = ((size_t)M * N + (csb ? N : 0)) * splits;
    float *A = nullptr, *B = nullptr, *C = nullptr, *bias = nullptr;
    SERT_TRY(dmalloc(&A, na)); SERT_TRY(dmalloc(&B, nb)); SERT_TRY(dmalloc(&C, nc)); SERT_TRY(dmalloc(&bias, (size_t)N));
    std::vector<float> h(std::max(std::max(na, nb), (size_t)N));
    uint32_t x = 12345u;
    auto fill = [&](float* d, size_t n) {
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) * (1.0f / 8388608.0f)) - 1.0f; }
        return hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    };
    SERT_HIP(fill(A, na)); SERT_HIP(fill(B, nb)); SERT_HIP(fill(bias, (size_t)N));
    const int lda = ta ? M : K, ldb = tb ? K : N;
#ifdef SERT_VARIANTS
    const int big = (variant_knob("SERT_GEMM_BIG") && !ta && tb && gemm_big_ok(K, lda, ldb) && splits == 1) ? atoi(variant_knob("SERT_GEMM_BIG")) : 0;
#endif
    auto run = [&]() {
#ifdef SERT_VARIANTS
        if (big) { launch_gemm_big_nt(s, A, B, C, M, N, K, lda, ldb, N, big == 2); return; }
#endif
#define SERT_BG(TA, TB, E) launch_gemm<TA, TB, E>(s, A, B, C, bias, M, N, K, lda, ldb, N, splits, kper, (size_t)M * N)
        if (!ta && !tb) { if (epi == 2) SERT_BG(false, false, EPI_BIAS_TANH); else if (epi == 1) SERT_BG(false, false, EPI_BIAS); else SERT_BG(false, false, EPI_STORE); }
        else if (csb) launch_gemm<true, false, EPI_STORE, true>(s, A, B, C, bias, M, N, K, lda, ldb, N, splits, kper, (size_t)M * N + N);
        else if (ta && !tb) SERT_BG(true, false, EPI_STORE);
        else if (!ta && tb) SERT_BG(false, true, EPI_STORE);
        else SERT_BG(true, true, EPI_STORE);
#undef SERT_BG
    };
    hipEvent_t e0, e1;
    SERT_HIP(hipEventCreate(&e0)); SERT_HIP(hipEventCreate(&e1));
    run(); run();
    SERT_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) run();
    SERT_HIP(hipEventRecord(e1, s));
    SERT_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    SERT_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1000.0 * ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(bias);
    (void)hipStreamDestroy(s);
    return 0;
}

// C = epi(op(A).op(B)) for host arrays, through launch_gemm -- i.e. through whichever kernel a shape is routed to in a
// training step (tests/test_gpu_gemm.py pins every kernel of gemm.h / gemm_stream.h against float64 this way).
int sert_debug_gemm(int device, int ta, int tb, int epi, int M, int N, int K, const float* A, const float* B,
                    const float* bias, float* C) {
    refresh_gemm_choice();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 2 || (epi && !bias)) SERT_FAIL("bad argument");
    SERT_HIP(hipSetDevice(device));
    hipStream_t s;
    SERT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t na = (size_t)M * K, nb = (size_t)K * N, nc = (size_t)M * N;
    float *dA = nullptr, *dB = nullptr, *dC = nullptr, *dbias = nullptr;
    int rc = 0;
    auto body = [&]() -> int {
        SERT_TRY(dmalloc(&dA, na)); SERT_TRY(dmalloc(&dB, nb)); SERT_TRY(dmalloc(&dC, nc)); SERT_TRY(dmalloc(&dbias, (size_t)N));
        SERT_HIP(hipMemcpyAsync(dA, A, na * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemcpyAsync(dB, B, nb * sizeof(float), hipMemcpyHostToDevice, s));
        if (bias) SERT_HIP(hipMemcpyAsync(dbias, bias, (size_t)N * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemsetAsync(dC, 0xff, nc * sizeof(float), s));      // (NaNs wherever the kernel does not write)
        const int lda = ta ? M : K, ldb = tb ? K : N;
#define SERT_DG(TA, TB, E) launch_gemm<TA, TB, E>(s, dA, dB, dC, dbias, M, N, K, lda, ldb, N)
        if (!ta && !tb) { if (epi == 2) SERT_DG(false, false, EPI_BIAS_TANH); else if (epi == 1) SERT_DG(false, false, EPI_BIAS); else SERT_DG(false, false, EPI_STORE); }
        else if (!ta && tb) { if (epi == 2) SERT_DG(false, true, EPI_BIAS_TANH); else if (epi == 1) SERT_DG(false, true, EPI_BIAS); else SERT_DG(false, true, EPI_STORE); }
        else if (ta && !tb) SERT_DG(true, false, EPI_STORE);
        else SERT_DG(true, true, EPI_STORE);
#undef SERT_DG
        SERT_HIP(hipGetLastError());
        SERT_HIP(hipMemcpyAsync(C, dC, nc * sizeof(float), hipMemcpyDeviceToHost, s));
        SERT_HIP(hipStreamSynchronize(s));
        return 0;
    };
    rc = body();
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dbias);
    (void)hipStreamDestroy(s);
    return rc;
}

// out (M * N + N) = A^T.B (A (K, M), B (K, N) host arrays) followed by the column sums of B, through the split-K launch
// + order-fixed combine the projection's dW / db take in a training step (tests/test_gpu_gemm.py)
int sert_debug_gemm_splitk(int device, int M, int N, int K, int splits, const float* A, const float* B, float* out) {
    refresh_gemm_choice();
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || splits <= 0) SERT_FAIL("bad argument");
    SERT_HIP(hipSetDevice(device));
    hipStream_t s;
    SERT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int kper = (int)round_up(cdiv(K, splits), GK);
    splits = cdiv(K, kper);
    const size_t na = (size_t)K * M, nb = (size_t)K * N, stride = (size_t)M * N + N;
    float *dA = nullptr, *dB = nullptr, *dP = nullptr, *dO = nullptr;
    auto body = [&]() -> int {
        SERT_TRY(dmalloc(&dA, na)); SERT_TRY(dmalloc(&dB, nb)); SERT_TRY(dmalloc(&dP, stride * splits)); SERT_TRY(dmalloc(&dO, stride));
        SERT_HIP(hipMemcpyAsync(dA, A, na * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemcpyAsync(dB, B, nb * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemsetAsync(dP, 0xff, stride * splits * sizeof(float), s));
        launch_gemm<true, false, EPI_STORE, true>(s, dA, dB, dP, nullptr, M, N, K, M, N, N, splits, kper, stride);
        launch_reduce_partials(s, dP, splits, stride, stride, dO, stride, dO);
        SERT_HIP(hipGetLastError());
        SERT_HIP(hipMemcpyAsync(out, dO, stride * sizeof(float), hipMemcpyDeviceToHost, s));
        SERT_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dP); (void)hipFree(dO);
    (void)hipStreamDestroy(s);
    return rc;
}

// C (M, N) = A . op(B) over a LONG K cut into `splits` k ranges (partial slabs + order-fixed combine): the form the loglinear
// dG = dZ.W^T takes over 100 000 entities (gemm_long_k).  A (M, K), B (K, N) or (N, K) if tb: host arrays.
int sert_debug_gemm_longk(int device, int tb, int M, int N, int K, int splits, const float* A, const float* B, float* C) {
    refresh_gemm_choice();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || splits <= 0) SERT_FAIL("bad argument");
    SERT_HIP(hipSetDevice(device));
    hipStream_t s;
    SERT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int kper = (int)round_up(cdiv(K, splits), GK);
    splits = cdiv(K, kper);
    const size_t na = (size_t)M * K, nb = (size_t)K * N, mn = (size_t)M * N;
    float *dA = nullptr, *dB = nullptr, *dP = nullptr, *dO = nullptr;
    auto body = [&]() -> int {
        SERT_TRY(dmalloc(&dA, na)); SERT_TRY(dmalloc(&dB, nb)); SERT_TRY(dmalloc(&dP, mn * splits)); SERT_TRY(dmalloc(&dO, mn));
        SERT_HIP(hipMemcpyAsync(dA, A, na * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemcpyAsync(dB, B, nb * sizeof(float), hipMemcpyHostToDevice, s));
        SERT_HIP(hipMemsetAsync(dP, 0xff, mn * splits * sizeof(float), s));
        if (tb) launch_gemm<false, true, EPI_STORE>(s, dA, dB, dP, nullptr, M, N, K, K, K, N, splits, kper, mn);
        else    launch_gemm<false, false, EPI_STORE>(s, dA, dB, dP, nullptr, M, N, K, K, N, N, splits, kper, mn);
        launch_reduce_partials(s, dP, splits, mn, mn, dO, mn, dO);
        SERT_HIP(hipGetLastError());
        SERT_HIP(hipMemcpyAsync(C, dO, mn * sizeof(float), hipMemcpyDeviceToHost, s));
        SERT_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dP); (void)hipFree(dO);
    (void)hipStreamDestroy(s);
    return rc;
}

int sert_bench_memory(int device, int kind, size_t bytes, size_t table_bytes, int row_bytes, int window,
                      size_t gap_bytes, int blocks, int iters, double* avg_us) {
    if (!avg_us || iters <= 0 || bytes < 4096) SERT_FAIL("bad argument");
    SERT_HIP(hipSetDevice(device));
    hipStream_t s;
    SERT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<void*> owned;
    auto alloc = [&](size_t b, void** out) -> int {
        SERT_HIP(hipMalloc(out, b));
        owned.push_back(*out);
        return 0;
    };
    auto cleanup = [&]() {
        for (void* p : owned) (void)hipFree(p);
        (void)hipStreamDestroy(s);
    };
    const size_t n = (bytes / 16) * 4;   // floats per array (multiple of 4)
    std::function<void()> run;
    int rc = 0;
    if (kind == SERT_MEMBENCH_COPY || kind == SERT_MEMBENCH_READ) {
        float *a = nullptr, *b = nullptr;
        if ((rc = alloc(n * 4, (void**)&a)) || (rc = alloc(kind == SERT_MEMBENCH_COPY ? n * 4 : 65536 * 4, (void**)&b))) { cleanup(); return rc; }
        hipLaunchKernelGGL(mb_fill_f32, dim3(2048), dim3(256), 0, s, a, n, 1.0f);
        const int nb = blocks > 0 ? blocks : 4096;
        if (kind == SERT_MEMBENCH_COPY)
            run = [=]() { hipLaunchKernelGGL(mb_stream_copy, dim3(nb), dim3(256), 0, s, (const float4*)a, (float4*)b, n / 4); };
        else
            run = [=]() { hipLaunchKernelGGL(mb_stream_read, dim3(nb), dim3(256), 0, s, (const float4*)a, n / 4, b); };
    } else if (kind == SERT_MEMBENCH_GATHER) {
        if (row_bytes < 16 || row_bytes % 16 || window < 1 || table_bytes < (size_t)row_bytes) { cleanup(); SERT_FAIL("bad gather shape"); }
        const int d = row_bytes / 4;
        const size_t rows = table_bytes / (size_t)row_bytes;
        const size_t B = std::max<size_t>(1, n / d);
        if (rows >= ((size_t)1 << 32) || B >= ((size_t)1 << 31)) { cleanup(); SERT_FAIL("gather shape too large"); }
        float *tab = nullptr, *out = nullptr; uint32_t* ids = nullptr;
        if ((rc = alloc(rows * row_bytes, (void**)&tab)) || (rc = alloc(B * row_bytes, (void**)&out)) ||
            (rc = alloc(B * window * 4, (void**)&ids))) { cleanup(); return rc; }
        hipLaunchKernelGGL(mb_fill_f32, dim3(2048), dim3(256), 0, s, tab, rows * d, 1.0f);
        hipLaunchKernelGGL(mb_fill_ids, dim3(2048), dim3(256), 0, s, ids, B * window, (uint32_t)rows, 17u);
        const int grid = grid_for((int64_t)B * d / 4, 256, 1 << 20);
        run = [=]() { hipLaunchKernelGGL((vs_gather_mean<uint32_t, 4>), dim3(grid), dim3(256), 0, s, (const uint32_t*)ids, (const float*)tab, out, (int)B, window, d); };
    } else if (kind == SERT_MEMBENCH_OPTIMIZER) {
        float* arr[4] = {nullptr, nullptr, nullptr, nullptr};
        if (gap_bytes == (size_t)-1) {   // four allocations of their own, as the model holds them
            for (int k = 0; k < 4; ++k) if ((rc = alloc(n * 4, (void**)&arr[k]))) { cleanup(); return rc; }
        } else {
            if (gap_bytes % 16) { cleanup(); SERT_FAIL("gap must be a multiple of 16 bytes"); }
            char* base = nullptr;
            if ((rc = alloc(4 * (n * 4 + gap_bytes), (void**)&base))) { cleanup(); return rc; }
            for (int k = 0; k < 4; ++k) arr[k] = (float*)(base + (size_t)k * (n * 4 + gap_bytes));
        }
        float* sq = nullptr;
        if ((rc = alloc((size_t)8 * kOptBlocks * 4, (void**)&sq))) { cleanup(); return rc; }
        hipLaunchKernelGGL(mb_fill_f32, dim3(2048), dim3(256), 0, s, arr[0], n, 0.01f);
        hipLaunchKernelGGL(mb_fill_f32, dim3(2048), dim3(256), 0, s, arr[1], n, 1e-4f);
        SERT_HIP(hipMemsetAsync(arr[2], 0, n * 4, s));
        SERT_HIP(hipMemsetAsync(arr[3], 0, n * 4, s));
        const int nb = blocks > 0 ? std::min(blocks, 8 * kOptBlocks) : 2 * kOptBlocks;
        const AdamArgs aa{1e-7f, 1e-3f, 0.9f, 0.999f, 1e-8f};
        float *p = arr[0], *g = arr[1], *m1 = arr[2], *v1 = arr[3];
        run = [=]() { hipLaunchKernelGGL((adam_l2<false>), dim3(nb), dim3(256), 0, s, p, g, m1, v1, n, aa, sq, (const uint32_t*)nullptr, 1u, (int)kRowsAll); };
    } else {
        cleanup();
        SERT_FAIL("unknown membench kind");
    }
    hipEvent_t e0, e1;
    SERT_HIP(hipEventCreate(&e0)); SERT_HIP(hipEventCreate(&e1));
    run(); run();
    SERT_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) run();
    SERT_HIP(hipEventRecord(e1, s));
    const hipError_t se = hipStreamSynchronize(s);
    float ms = 0.f;
    if (se == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    *avg_us = 1000.0 * ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    cleanup();
    SERT_HIP(se);
    SERT_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"


#ifdef SR_TIMELINE
extern "C" int sert_debug_read(void* out, size_t bytes) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sert::sr_dbg), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
