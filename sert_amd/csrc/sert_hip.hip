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
static int settle_tail(sert_model* m);
static void invalidate_speculation(sert_model* m) {
    discard_run_ahead(m);
    (void)settle_tail(m);
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

#include "host/lazy_segsum.inc"

#include "host/data_parallel.inc"

#include "host/step_vectorspace.inc"

#include "host/step_softmax_loglinear.inc"

#include "host/optimizer_and_loss.inc"

}  // namespace sert

using namespace sert;

// =============================== C ABI ===========================================
extern "C" {

#include "host/api_model.inc"

#include "host/api_data_train.inc"

#include "host/api_scorer.inc"

#include "host/api_comm.inc"

#include "host/api_debug.inc"

}  // extern "C"


#ifdef SR_TIMELINE
extern "C" int sert_debug_read(void* out, size_t bytes) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sert::sr_dbg), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
