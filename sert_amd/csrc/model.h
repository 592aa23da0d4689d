// Device-side model state for libsert_hip.so.
#pragma once
#include <vector>
#include <string>
#include "common.h"
#include "word_index.h"
#include "kernels_xchg.h"
#include "../../include/sert_hip.h"
#include "../../include/sert_hip_debug.h"   // (test hooks + micro-benchmarks: declared apart from the boundary)

namespace sert {

struct DataSplit {
    int64_t N = 0;
    void* x = nullptr;          // (N, n) ids, id_bytes wide
    int32_t* y = nullptr;       // (N,) int labels, or null
    int64_t* csr_indptr = nullptr;   // (N+1)
    int32_t* csr_indices = nullptr;  // (nnz)
    float* csr_data = nullptr;       // (nnz)
    int64_t nnz = 0;
    int64_t max_labels_per_row = 0;
    float* w = nullptr;         // (N,) instance weights (train split)
    float* labfix = nullptr;    // loglinear streaming loss: per label entry Q_e dQ_e (N or nnz)
    // inverted index word -> rows of every complete batch (train split only)
    int32_t* idx_rows = nullptr;
    int4* idx_items = nullptr;
    int4* idx_heavy = nullptr;       // heavy words of three-level trees (word_index.h: heavy_off)
    int32_t* idx_bundles = nullptr;  // level-0 bundles (word_index.h: bundle_off)
    int32_t* idx_uwords = nullptr;   // loglinear: distinct words of every batch (sorted)
    int32_t* idx_slots = nullptr;    // loglinear: per token position, rank of its word among them
    int32_t* idx_rows_div = nullptr; // loglinear: idx_rows / n (batch row of every level-0 entry)
    uint32_t* idx_touched_bits = nullptr;   // per batch: bit w set iff word w occurs in it (word_index.h)
    uint4* idx_dense_counts = nullptr;      // per batch and row: occurrence counts of the batch's dense heavy words
    int32_t* idx_dense_words = nullptr;     // per batch: their word ids (kHeavyMax slots)
    uint8_t* idx_tok_slot = nullptr;        // (vectorspace) per batch and token position: dense slot of its word or 255
    std::vector<int32_t> dense_cnt_of;      // per batch: number of dense words (host copy for the forward's gather)
    int64_t bit_words = 0;                  // 32-bit words per batch in idx_touched_bits
    std::vector<BatchIndex> idx_batches;
};

// Timed kernel groups (HIP events on the model's stream).
enum TimingGroup {
    TG_GATHER = 0,    // vs_gather_mean / ll_gather_rows                      (1 launch)
    TG_GEMM_FWD,      // gemm_f32_mfma: projection / logits                   (1 launch)
    TG_LOSS,          // vs_nce / ll_fused_row (or ll_softmax_rows+ll_window)  (1-2 launches)
    TG_SORT,          // csort_hist + csort_scan_bins + csort_scatter per digit
    TG_EGRAD,         // egrad_chunk_reduce                                   (1 launch)
    TG_EFIX,          // egrad_fixup                                          (1 launch)
    TG_GEMM_DW,       // gemm_f32_mfma<TN, split-K, +column sums>             (1 launch)
    TG_SPLITK,        // reduce_partials                                      (1 launch)
    TG_GEMM_DX,       // gemm_f32_mfma<NT>: dh / dG                           (1 launch)
    TG_SCATTER,       // segsum_rows, one launch per tree level
    TG_ALLREDUCE,     // RCCL all-reduce of the small (replicated) tensors' gradients + loss scalars
    TG_REDUCE_SCATTER,// RCCL reduce-scatter of the big tensors' gradients (ZeRO-1 ownership)
    TG_ALLGATHER,     // RCCL all-gather of the big tensors' updated parameters
    TG_OPT_WORD,      // adam_l2 / adadelta_l2 on the word table R_w          (1 launch)
    TG_OPTIMIZER,     // the other tensors (R_e, W, b)                        (2-3 launches)
    TG_FINALIZE,      // loss reduction
    TG_COUNT
};

struct Timing {
    bool enabled = false;
    hipEvent_t ev[TG_COUNT][2] = {};
    bool created = false;
    bool used[TG_COUNT] = {};
    double total_us[TG_COUNT] = {};
    int64_t samples[TG_COUNT] = {};
};

// sert_timing_enable(m, 2) (common.h: InStepHook): event pairs of the launches in flight, harvested when the ring is full
// and when the averages are read
struct InStep {
    static constexpr int kRing = 1024;
    bool on = false;
    bool created = false;
    hipEvent_t ev[kRing][2] = {};
    int group[kRing] = {};
    int64_t head = 0, tail = 0;        // pairs [head, tail) are pending
    int cur_group = -1;                // the timing group the launching thread is inside (ScopedTimer)
    double total_us[TG_COUNT] = {};
    int64_t launches[TG_COUNT] = {};
    int64_t steps = 0;
};

}  // namespace sert

struct sert_model {
    sert_config cfg;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // side stream: the entity-gradient chain runs beside the GEMMs
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    hipStream_t stream3 = nullptr;   // dW + split-K combine, beside dX/segsum and the entity chain
    // step prologue (zeroing, negative sampling) and the small-tensor optimiser run on
    // stream2 beside the main chain; these events order them
    hipEvent_t ev_step_done = nullptr, ev_neg = nullptr, ev_opt_fork = nullptr, ev_small = nullptr;
    hipEvent_t ev_loss = nullptr;    // the step's loss has been copied out
    hipEvent_t ev_dense = nullptr;   // dW, db and the loss partials are complete (main stream)
    bool step_done_pending = false;  // the previous step ended without recording ev_step_done
    bool re_in_parts = false;        // this step: dR_e is still the row groups' partial tables (summed by the optimiser)
    bool fork_bound = false;         // ev_fork rides on the NCE kernel's completion signal (no record needed)
    bool egrad_ranges = false;       // SERT_EGRAD_RANGES=1 at sert_create: the one-launch range kernel for few pairs over a mid-size table (opt-in)
    bool egrad_force_sort = false;   // SERT_EGRAD_SORT=1 at sert_create: the sorted entity-gradient path whatever the shape
    bool events_device_scope = false;  // the intra-model events carry hipEventDisableSystemFence (no communicator; sert_hip.hip: create_intra_events)
    bool lazy_join = false;          // this step: the main stream never waits for the entity chain
    int num_cus = 256;               // compute units of the device (persistent launches: one workgroup per CU)
    bool proj_fused = false;         // gather + mean-pool + projection in one launch where the shape allows (kernels_proj.h; opt-in, SERT_PROJ_FUSED=1)
    bool ll_dw_side = false;         // loglinear, this step: dW, db and their combine were issued on the side stream
    bool dw_side_first = false;      // this step: dW / db came from the side stream, FIRST in its chain (ev_dense marks them)
    bool dp_late_join = false;       // data parallel, asynchronous communicator: the side stream (entity chain, dW, db, loss sum) is
                                     // joined by the COMMUNICATION stream in front of the small all-reduce, not by the main stream
    bool side_heavy = false;         // this step: entity chain, entity-table optimiser and dW on the side stream (big R_e)
    // side-heavy schedule: the entity table's update is DEFERRED past the step's tail -- it only has to land
    // before the next reader of R_e (the next loss kernel); the sums of squares the tail needs were left by
    // the previous step's launch (re_sq[k]: partials of the updated table, for optimiser step re_sq_for[k])
    hipEvent_t ev_re = nullptr;
    bool re_pending = false;
    bool tail_early = false, w_early_pending = false;   // (knock-out SERT_KO_TAIL_EARLY: see sert_hip.hip)
    hipStream_t tail_stream = nullptr;   // the step's tail on a queue of its own (optimizer_and_loss: tail_queue)
    hipEvent_t ev_tail_go = nullptr, ev_tail_done = nullptr;
    bool tail_pending = false;
    int tail_queue_min_batch = 0;
    bool w_pending = false;          // W, b were updated on the side stream too (same event): the next projection waits
    float* re_sq = nullptr;          // [2][2 * kOptBlocks]
    int64_t re_sq_for[2] = {-1, -1};
    int n_loss_partials = 0;
    bool loss_from_rows = false;     // this step: the loss finalisation reads the per-row losses directly (few rows)
    int nce_loss_partials = 0;       // > 0: vs_nce wrote this many per-workgroup loss partials into red_loss
    // SERT_STREAMS: 1 = everything on the main stream (0.423 ms/step at C2), 2 = + the entity
    // chain, the step prologue and the small-tensor optimiser on a side stream (0.396),
    // 3 = + dW on a third (0.403: every cross-queue dependency costs 15-25 us of idle GPU)
    int nstreams = 2;
    int64_t hint_next = -1;        // sert_hint_next_batch
    bool neg_side_ready = false;   // this step's negatives were drawn on the side stream during the previous step
    bool sort_early = false;       // this step's entity keys were sorted in front of the fork (sert_hip.hip: vs_backward, early_sort)
    bool bucket_early = false;     // this step's egrad_bucket ran in front of the fork (sert_hip.hip: vs_backward)
    // lazy dense update of the word table (kernels_opt.h: dense_update_lazy)
    int32_t* rw_last[2] = {nullptr, nullptr};   // per word row: updates applied to its stored (p, state0, state1)
    int rw_last_cur = 0;           // which of the two holds the current values
    bool rw_stale = false;         // some rows are behind m->step (rw_last[rw_last_cur] says by how much)
    int64_t rw_ready_batch = -1;   // while stale: the training batch whose rows ARE current (the hinted one)
    int64_t lazy_next = -1;        // the batch the caller announced to follow the one being trained
    float cur_touched_frac = 1.f;  // distinct words of the batch being trained / vocabulary
    // dense_update_skip (kernels_opt.h): per-row shares of sum(p^2) along the zero-gradient trajectory, [kLazyK][stride]
    float* rw_pred = nullptr;
    unsigned rw_pred_stride = 0;
    bool rw_pred_ok = false;       // every row that is behind has its predictions up to (not including) update rw_pred_T
    int64_t rw_pred_T = 0;         // the next update that reads every row
    bool lazy_skip = true;         // SERT_LAZY_SKIP=0: dense_update_lazy (reads every row every step)
    float lazy_max = 0.5f;         // SERT_LAZY_MAX: largest touched fraction of a batch whose word-table update is lazy
    // launches of the word-table update by form since sert_create (host counters, read by sert_debug_update_counts -- the
    // tests assert through them that every template shape of dense_update_skip was in front of the oracle):
    // [0] dense (adam_l2 / adadelta_l2), [1] dense_update_lazy, [2..7] dense_update_skip <32,1> <64,1> <32,3> <64,2> <64,3>
    // <64,4>, [8] of those the passes that read and write every row, [9] the sparse ones
    int64_t upd_counts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t projected_batch = -1;  // training batch whose forward projection already sits in H/T
    // hinted single-GPU steps go further: the whole forward + backward runs ahead
    int64_t spec_fb_batch = -1;    // forward + backward of this batch already ran (gradients ready) ...
    int64_t spec_fb_step = -1;     // ... as optimiser step spec_fb_step
    unsigned loss_seq = 0;        // sequence number the final kernel publishes beside the loss
    float* h_loss_dev = nullptr;  // device address of the pinned h_loss block

    // shapes
    size_t n_rw = 0, n_re = 0, n_w = 0, n_b = 0;

    // parameters + optimiser state
    float *rw = nullptr, *re = nullptr, *W = nullptr, *b = nullptr;
    float *s0_rw = nullptr, *s0_re = nullptr, *s0_w = nullptr, *s0_b = nullptr;
    float *s1_rw = nullptr, *s1_re = nullptr, *s1_w = nullptr, *s1_b = nullptr;

    // gradients: ONE flat allocation [g_rw | g_re | g_w | g_b | loss_sum(1) | pad].
    // Data-parallel exchange = two all-reduces over it: the word-table slice as soon
    // as it is complete (overlapping the rest of the backward), then the remainder.
    float* gflat = nullptr;
    size_t gflat_count = 0;
    float *g_re = nullptr, *g_rw = nullptr, *g_w = nullptr, *g_b = nullptr, *g_loss = nullptr;
    size_t gflat_alloc = 0;       // allocation incl. the per-step-zeroed int tail below
    int32_t *run_start = nullptr, *run_end = nullptr;   // (V_e) sorted-run bounds per entity

    // per-batch activations
    float *H = nullptr, *T = nullptr, *DA = nullptr, *DH = nullptr, *rowloss = nullptr;
    float* T_alt = nullptr;          // the projection alternates between two buffers (vs_project): the entity chain of the previous
                                     // step may still read its rows on the side stream when the next projection is written
    float* DH2 = nullptr;         // full-softmax variant: p = clip(t)  (B, d_e)
    // full-softmax variant: the logits exist for fs_tile rows at a time (== batch_size: the whole batch)
    int fs_tile = 0;
    int32_t* neg = nullptr;       // (B, z) device negatives
    // the NEXT training step's negatives, drawn at the end of this step on the side stream (they
    // depend on (seed, step, row) only): the sampler leaves the next step's critical path
    int32_t* neg_alt = nullptr;
    int64_t neg_alt_step = -1;    // optimiser step the buffer was drawn for (-1: none)
    int64_t* neg_stage = nullptr; // (B, z) int64 staging for host-supplied negatives
    // entity-gradient machinery (kernels_egrad.h), all (B*(1+z)) long
    int32_t *cand = nullptr, *cand_sorted = nullptr, *pair_sorted = nullptr;
    int32_t* cand_early = nullptr;   // the same keys built from labels + negatives in front of the loss kernel (early_sort)
    float* coef = nullptr;
    float *ehead = nullptr, *etail = nullptr;  // (chunks, d_e) carries
    // small entity vocabularies: the sort-free path (egrad_lds, kernels_egrad.h)
    float* epart = nullptr;       // (groups, V_e, d_e) per-row-group partial entity gradients
    int32_t *eg_entries = nullptr, *eg_offs = nullptr;   // pairs bucketed by entity range per sub-group; their offsets
    int eg_sub_rows = 0, eg_num_sub = 0, eg_subs_per_group = 0, eg_groups = 0, eg_ranges = 0, eg_er_shift = 0;
    int32_t *sort_hist = nullptr, *sort_bin_total = nullptr;   // counting-sort scratch
    int32_t *sort_k_tmp = nullptr, *sort_v_tmp = nullptr;       // ping-pong (only if > 11 key bits)
    int sort_bits = 1;
    // loglinear activations
    float *G = nullptr;           // (B*n, d) gathered rows
    float *Z = nullptr;           // (B*n, V_e) logits -> probabilities -> dZ
    float *J = nullptr;           // (B, V_e) window log-product -> dJ
    float *DG = nullptr;          // (B*n, d)

    // scratch
    float* wpart = nullptr;       // segmented-reduce partial rows (word gradient tree)
    float* hpart = nullptr;       // dense heavy words: per row-block partial rows [blocks][kHeavyMax][d_w]
    size_t wpart_rows = 0;
    float* part = nullptr;        // split-K partials
    size_t part_count = 0;
    // loglinear streaming loss (kernels_ll.h, ll_s_*): per (row, segment) partials
    float2* ll_tokstat = nullptr; float* ll_lse = nullptr; float2* ll_jstat = nullptr;
    float4* ll_rowinfo = nullptr; float* ll_rpart = nullptr; float* ll_r = nullptr;
    // single GPU: the word-gradient table is neither zeroed nor read where no token of the
    // batch points (static per-batch row bitmaps, DataSplit::idx_touched_bits) ...
    bool use_touched = false;
    // ... and the optimiser of those untouched rows -- rows the batch's own forward never reads --
    // is issued at the START of the step on its own stream, beside forward and backward
    // (opt-in, SERT_ADAM_SPLIT=1: measured SLOWER than one launch behind the backward, see sert_hip.hip)
    hipStream_t stream4 = nullptr;
    hipEvent_t ev_word_opt = nullptr;   // the touched rows' update of the previous step was issued (main stream)
    hipEvent_t ev_early = nullptr;      // this step's untouched-row update is complete (stream4)
    bool early_issued = false;          // this step's untouched-row launch is in flight
    int early_sq = 0;                   // ... and wrote this many sum-of-squares partials to red_sq[0..)

    // loglinear, logits per DISTINCT word of the batch (duplicate tokens share a row):
    float* Zu = nullptr;          // (U, V_e) logits
    float* dZu = nullptr;         // (U, V_e) per-word sums of dL/dZ
    size_t zu_rows = 0;
    float* zpart = nullptr;       // V_e-wide partial rows of the per-word sum tree
    size_t zpart_rows = 0;
    bool ll_dedup = false;        // this step ran on the distinct-word table
    float* ll_rsum = nullptr;     // (U) per-word sums of r_ik
    int ll_U = 0;
    float* skbuf = nullptr;       // split-K partials of the long-K dX GEMMs (grown on demand)
    size_t skbuf_count = 0;
    // single-GPU vectorspace step: split-K combine + W, b update + loss finalisation as one launch
    // (kernels_opt.h: vs_tail).  tail_splits > 0: this step's dW / db still sit in `part` as that
    // many partial slabs, tail_stride elements apart
    const float* tail_part = nullptr;   // where the tail finds dW | db: the split-K slabs, or their sums (combined on the side stream)
    unsigned long long* tail_blk = nullptr;
    unsigned tail_launch_seq = 0;
    int tail_splits = 0;
    size_t tail_stride = 0;
    float* red_loss = nullptr;    // loss partials [kOptBlocks]
    float* red_sq = nullptr;      // sumsq partials [3 * kOptBlocks]
    float* d_loss = nullptr;      // [3] loss, data term, reg term (device)
    float* h_loss = nullptr;      // pinned host mirror
    float* d_losses = nullptr;    // multi-step loss ring
    int64_t d_losses_cap = 0;

    // grow-only scratch of the predict functions (EmbeddingMapper calls predict_fn once per query:
    // no hipMalloc / hipFree per call, nothing to leak on an error path)
    float *pred_a = nullptr, *pred_b = nullptr;   // inputs / gathered rows; outputs
    void* pred_ids = nullptr;
    size_t pred_a_cap = 0, pred_b_cap = 0, pred_ids_cap = 0;

    sert::DataSplit split[2];

    int64_t step = 0;             // optimiser step counter t (Adam) / training sampler position
    int64_t eval_draws = 0;       // evaluation sampler position

    // ---- the four parameter tensors as the optimiser sees them: [R_w, R_e, W, b] ----
    // A "big" tensor (the word table always; R_e / W beyond 4 M elements) is updated by its own
    // streaming launch.  Data parallel, big tensors are owned ZeRO-1 style: the padded tensor
    // (pt_pad elements) is cut into `chunks` slabs of world * pt_sc elements and rank r owns
    // [c*world*sc + r*sc, + sc) of every slab c -- reduce-scatter of the gradient slab, the
    // optimiser on the owned piece, all-gather of the parameter slab, slab after slab.  The
    // optimiser state of a sharded tensor exists for the owned pieces only (chunks * sc
    // elements, slab-major) -- 1/world of the replicated state.
    size_t pt_pad[4] = {0, 0, 0, 0};   // allocated elements of p and g (>= n, multiple of 4)
    size_t pt_sc[4] = {0, 0, 0, 0};    // owned elements per slab (sharded tensors)
    bool pt_big[4] = {false, false, false, false};
    bool pt_sharded[4] = {false, false, false, false};
    size_t ar_split = 0;          // gflat[0, ar_split) = word-table gradient (incl. its padding)
    size_t rest_off = 0;          // gflat[rest_off, gflat_count) = replicated tensors' gradients + scalars
    float* g_sq = nullptr;        // scalar slot: sum of squares of the sharded tensors (all-reduced)
    float* sq_scratch = nullptr;  // per-block partials nobody reads (sharded optimiser launches)

    // data parallel
    int rank = 0, world = 1;
    void* comm = nullptr;         // ncclComm_t
    bool comm_dead = false;       // communicator destroyed: the (sharded) model can no longer train
    // host-mediated exchange (sert_comm_init_host): verification transport.  ONE primitive -- an
    // all-to-all of float segments between the ranks -- carries every collective the step needs, each
    // rank receiving exactly its pieces (reduce-scatter, all-gather, the all-to-alls of the row exchange,
    // the small all-reduce), so the piece / padding / slab indexing is what a multi-rank test exercises
    sert_alltoall_fn host_ar = nullptr;
    void* host_ar_user = nullptr;
    float *host_send = nullptr, *host_recv = nullptr;   // pinned staging
    size_t host_send_cap = 0, host_recv_cap = 0;

    // ---- the word table exchanged BY ROWS (kernels_xchg.h) ----
    bool xr_mode = false;            // R_w is owned by rows (decided when the communicator is attached)
    bool xr_on = false;              // ... and the exchange lists of the uploaded training split exist
    int64_t xr_rows_per_rank = 0;
    sert::RowExchangeLists* xr = nullptr;      // host copy: per batch and peer counts, offsets
    int32_t *xr_serve = nullptr, *xr_fetch = nullptr, *xr_union = nullptr, *xr_ent = nullptr, *xr_ptr = nullptr;
    uint32_t* xr_ubits = nullptr;
    float *xr_send = nullptr, *xr_recv = nullptr;      // (max rows any batch moves) x d_w
    bool rw_full = true;             // every row of R_w held by this rank is current
    int64_t xr_fetched_batch = -1;   // rw_full == false: the rows this batch touches are current
    int64_t xr_batch = -1;           // batch of the training step in flight
    hipEvent_t ev_params_ready = nullptr, ev_word_updated = nullptr;
    // bytes this rank sent + received through collectives, and steps counted (sert_comm_stats)
    double comm_bytes_moved = 0.0;
    int64_t comm_steps = 0;
    hipStream_t comm_stream = nullptr;          // all collectives are issued here, in one fixed order
    hipEvent_t ev_rest_ready = nullptr, ev_ar_done = nullptr;
    // the exchange of a big tensor is cut into ar_chunks slabs so that the optimiser of
    // slab c runs while slab c+1 is still on the links
    static constexpr int kMaxArChunks = 16;
    int ar_chunks = 1;
    hipEvent_t ev_grad_ready[4] = {};                 // tensor i's gradient is complete (producer stream)
    hipEvent_t ev_rs_done[4][kMaxArChunks] = {};      // slab c of tensor i reduce-scattered (comm stream)
    hipEvent_t ev_opt_done[4][kMaxArChunks] = {};     // owned piece of slab c updated (main stream)
    hipEvent_t ev_ag_done = nullptr;                  // every parameter slab of this step all-gathered
    bool rs_issued[4] = {false, false, false, false}; // this step's reduce-scatter of tensor i is in flight
    hipStream_t sq_stream = nullptr;                  // (the side stream the shard sum-of-squares runs on)

    sert::Timing timing;
    sert::InStep instep;
};

struct sert_scorer {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t V = 0;
    int dim = 0;
    float* E = nullptr;      // (V, dim) L2-normalised entity table
    float* P = nullptr;      // (Q, dim) query projections
    float* S = nullptr;      // 2 x (QT, V) cosine slabs: query tiles alternate between two
                             // streams so the GEMM of tile t+1 overlaps the top-k of tile t
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    float* val = nullptr;    // (Q, k)
    int32_t* idx = nullptr;  // (Q, k)
    int64_t cap_q = 0, cap_qk = 0, cap_s = 0;
    // fused path (GEMM with filtering epilogue): sample cosines, thresholds, candidate lists
    float* Ss = nullptr; float* thr = nullptr;
    unsigned long long* cand = nullptr; unsigned char* cnt = nullptr;
    int* nflag = nullptr; int* flag_list = nullptr;
    float* Pc = nullptr; int32_t* idx_c = nullptr; float* val_c = nullptr;
    int64_t cap_ss = 0, cap_ft = 0, cap_cand = 0, cap_flag = 0, cap_c = 0, cap_ck = 0;
    // bf16 prefilter (kernels_score_bf16.h): bf16 copies of E and of the current query tile,
    // rows zero-padded to kp columns
    bool bf16 = false, bf16_demoted = false;
    int kp = 0;
    uint16_t* E16 = nullptr; uint16_t* P16 = nullptr;
    int64_t cap_p16 = 0;
};
