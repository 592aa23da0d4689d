/*
 * sert_hip.h -- C ABI of libsert_hip.so, the MI355X (gfx950) execution engine
 * behind the sert.models / sert.inference Python surface.
 *
 * The reference (cvangysel/SERT) has no FFI: its hot path is the body of four
 * compiled Theano functions created in sert/models.py:530-608 (train_fn,
 * test_fn, validate_fn, predict_fn).  Each entry point below replaces one of
 * those functions or one Theano primitive they rely on; the file:line it
 * replaces is cited next to it.  The Python host (sert_amd/models.py) binds
 * these with ctypes; INTEGRATION.md shows the stub a SERT maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from sert_last_error() (thread-local, valid until the next call)
 *   - plain pointers and sizes only; host pointers are borrowed for the
 *     duration of the call; device memory is owned by the sert_model handle
 *   - one handle = one model replica on one HIP device with one HIP stream;
 *     a handle is not thread-safe
 *   - all floating point data is IEEE fp32 (floatX=float32, product-search.sh:95)
 */
#ifndef SERT_HIP_H
#define SERT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sert_model sert_model;

/* model kind: bin/train.py:21-24 */
enum {
    SERT_KIND_LOGLINEAR = 0,   /* sert.models.LanguageModel            (models.py:804) */
    SERT_KIND_VECTORSPACE = 1, /* sert.models.VectorSpaceLanguageModel (models.py:1024) */
    /* ADDITIVE, not in the reference (BASELINE.json configs[1] wording
     * "embed gather + MFMA projection + full softmax"): the vectorspace encoder
     * (gather, mean-pool, tanh projection) scored against ALL entities,
     * logits = clip(t).R_e^T, clipped softmax cross-entropy, dense L2, Adam. */
    SERT_KIND_VECTORSPACE_SOFTMAX = 2
};

/* data split: ModelBase.TRAIN / VALIDATE givens (models.py:482-522) */
enum { SERT_SPLIT_TRAIN = 0, SERT_SPLIT_VALIDATE = 1 };

/* tensors addressable through sert_get_tensor / sert_set_tensor */
enum {
    SERT_T_RW = 0,      /* word representations   (V_w, d_w)  models.py:167-171 */
    SERT_T_RE = 1,      /* entity representations (V_e, d_e)  models.py:940     */
    SERT_T_W = 2,       /* dense weights: VS (d_w, d_e) models.py:1057; LL (d_w, V_e) :846 */
    SERT_T_B = 3,       /* dense bias:    VS (d_e);  LL (V_e)                    */
    /* optimiser state, same shapes: Adam m / Adadelta accu  = STATE0,
     *                               Adam v / Adadelta delta = STATE1 */
    SERT_T_STATE0_RW = 4, SERT_T_STATE0_RE = 5, SERT_T_STATE0_W = 6, SERT_T_STATE0_B = 7,
    SERT_T_STATE1_RW = 8, SERT_T_STATE1_RE = 9, SERT_T_STATE1_W = 10, SERT_T_STATE1_B = 11,
    /* last-step gradients (data term + L2 term) and activations: readable only from a
     * model created with cfg.keep_grads (sert_get_tensor fails otherwise: without it these
     * buffers are scratch) */
    SERT_T_GRAD_RW = 12, SERT_T_GRAD_RE = 13, SERT_T_GRAD_W = 14, SERT_T_GRAD_B = 15,
    /* last-step activations (debug / parity): */
    SERT_T_ACT_H = 16,   /* VS mean-pooled window (B, d_w)      models.py:226  */
    SERT_T_ACT_T = 17,   /* VS tanh(hW+b)         (B, d_e)      models.py:1057 */
    SERT_T_ACT_DA = 18,  /* VS dL/d(hW+b)         (B, d_e)                     */
    SERT_T_ACT_DH = 19,  /* VS dL/dh              (B, d_w)                     */
    SERT_T_ACT_ROWLOSS = 20 /* per-instance loss  (B,)          models.py:1098, :292 */
};

typedef struct sert_config {
    uint32_t struct_size;     /* = sizeof(sert_config), checked */
    int32_t kind;             /* SERT_KIND_* */
    int32_t batch_size;       /* rows this replica processes per step (B_local) */
    int32_t global_batch_size;/* B of the model: divisor of the loss mean and of the
                                 L2 term lambda/(2B) (models.py:281-282, :773-791).
                                 == batch_size unless data-parallel */
    int32_t window_size;      /* n,   models.py:697 */
    int32_t vocab_size;       /* V_w, models.py:701 */
    int32_t num_entities;     /* V_e, models.py:924 / output_layer_size :809 */
    int32_t word_dim;         /* d_w, models.py:702 */
    int32_t entity_dim;       /* d_e, models.py:925 (vectorspace only) */
    int32_t num_negatives;    /* z,   models.py:961 (vectorspace only) */
    int32_t id_bytes;         /* width of token ids in x: 1, 2 or 4
                                 (np.min_scalar_type, bin/prepare.py:380) */
    int32_t device;           /* HIP device ordinal */
    int32_t keep_grads;       /* keep SERT_T_GRAD_* readable after a step (debug) */
    int32_t deterministic;    /* 1: order-fixed reductions everywhere */
    int32_t inference_only;   /* 1: parameters only (no optimiser state, gradients or
                                 activations); only sert_predict_* and tensor I/O work */
    float lambda_;            /* regularization_lambda, models.py:704 */
    /* optimiser hyper-parameters.  vectorspace: Adam (models.py:922)
     * lr, beta1, beta2, eps.  loglinear: Adadelta (models.py:820) lr, rho(beta1), eps. */
    float lr, beta1, beta2, eps;
    uint64_t seed;            /* device negative sampler (replaces RandomStreams
                                 seeding, models.py:958-959) */
} sert_config;

/* ---- lifetime ----------------------------------------------------------- */

/* Replaces model construction + theano.function compilation
 * (models.py:422-480, :530-608). Allocates parameters (zero), optimiser state,
 * gradient and activation buffers on cfg->device. */
int sert_create(const sert_config* cfg, sert_model** out);
int sert_destroy(sert_model* m);

const char* sert_last_error(void);

/* Human-readable device description ("gfx950 ... 256 CUs ..."); returns the
 * number of bytes written (excluding NUL) or <0 on error. */
int sert_device_info(int device, char* buf, size_t buflen);
/* Number of visible HIP devices, <0 on error. */
int sert_device_count(void);

/* ---- parameters and state ---------------------------------------------- */

/* theano.shared(...).set_value / get_value (models.py:183, :328-330, :945).
 * `count` must equal the tensor's element count. */
int sert_set_tensor(sert_model* m, int which, const float* host, size_t count);
int sert_get_tensor(sert_model* m, int which, float* host, size_t count);
/* element count of a tensor (0 if not present for this model kind) */
size_t sert_tensor_size(sert_model* m, int which);

/* Adam's shared step counter t (Lasagne adam t_prev); resume support. */
int sert_set_step(sert_model* m, int64_t t);
int64_t sert_get_step(sert_model* m);

/* Position of the EVALUATION negative stream (the reference keeps a second RandomStreams for
 * the evaluation loss, models.py:751 -> :1074; here: the odd Philox stream, advanced once per
 * evaluated batch).  With sert_set_step -- the training stream's position is the optimiser
 * step -- a resumed run draws what the uninterrupted run would have drawn. */
int sert_set_eval_draws(sert_model* m, int64_t n);
int64_t sert_get_eval_draws(sert_model* m);

/* The (B, z) negatives the device sampler draws for the training step taken when `position` optimiser updates have been
 * applied (evaluation = 0; position = sert_get_step before the step), or for the `position`-th evaluated batch
 * (evaluation = 1) -- what RandomStreams.choice (models.py:970-973) hands the reference's graph.  out (B, z) int64.  A pure
 * function of (seed, position, rank): no state of the model changes, so a caller can replay a device-sampled run with
 * explicit negatives (sert_train_batch's `negatives`) or feed the same ids to a CPU implementation. */
int sert_negatives_of_step(sert_model* m, int64_t position, int evaluation, int64_t* out);

/* ---- data set ---------------------------------------------------------- */

/* Replaces theano.shared(x/y/w) of the whole data set (models.py:470-480):
 * one H2D upload, batches are then slices [i*B, (i+1)*B) (models.py:322-326).
 *   x        (N, n) token ids, id_bytes wide, row-major
 *   y_int    (N,) int32 labels, or NULL when the labels are CSR
 *   csr_*    CSR label matrix (N, V_e) f32 (loglinear without --one_hot_classes,
 *            models.py:66-89 densifies it per batch); NULL when y_int given
 *   w        (N,) f32 instance weights, NULL = all ones (train split only)
 * In data-parallel mode every rank uploads the rows it owns of every global
 * batch (see sert_amd/distributed.py). */
int sert_upload_dataset(sert_model* m, int split, const void* x, const int32_t* y_int,
                        const int64_t* csr_indptr, const int32_t* csr_indices,
                        const float* csr_data, const float* w, int64_t num_instances);

/* ---- the hot path ------------------------------------------------------- */

/* train_fn(batch_index) (models.py:581-588): forward, backward, L2, optimiser
 * update on rows [batch_index*B, (batch_index+1)*B) of the train split.
 * *loss_out receives the training loss evaluated BEFORE the update.
 *   negatives  (B, z) int64 entity ids (vectorspace; models.py:970-973) or NULL
 *              to let the device sampler draw them (iid uniform, with
 *              replacement, keyed by (seed, step, global row, j)).
 * With a communicator attached (sert_comm_init) the gradients are summed over
 * ranks before the (replicated) update, and *loss_out is the global loss. */
int sert_train_batch(sert_model* m, int64_t batch_index, const int64_t* negatives,
                     float* loss_out);

/* Additive.  Announces the batch the call AFTER the next sert_train_batch will train.
 * The next sert_train_batch then lets that batch run ahead of the host: behind its own
 * step, and BEFORE it waits for its own loss, it enqueues the announced batch's forward
 * and backward (single GPU, device-drawn negatives; data parallel: the parameter-only
 * forward projection) -- everything that depends on parameters, data and the step
 * counter only and writes activations and gradient scratch only -- so the device does not
 * idle through the host round trip of the reference's per-batch loop
 * (sert/models.py:369-379: train_fn, isfinite check, next train_fn).  The announced step's
 * UPDATE (optimiser, loss) is never issued ahead: a step that raises on a non-finite loss
 * leaves the model exactly as without the hint.  A following call that does not match
 * (another batch, explicit negatives, an evaluation, new parameters or data) simply
 * recomputes.  next_batch_index < 0 clears the hint. */
int sert_hint_next_batch(sert_model* m, int64_t next_batch_index);

/* Same, for `count` batches back to back with no host synchronisation in
 * between (device sampler only).  losses_out[count]. */
int sert_train_batches(sert_model* m, const int64_t* batch_indices, int64_t count,
                       float* losses_out);

/* test_fn / validate_fn (models.py:593-608): unweighted, unregularised mean
 * loss of one batch of `split`; no parameter change. */
int sert_eval_batch(sert_model* m, int split, int64_t batch_index,
                    const int64_t* negatives, float* loss_out);

/* The same for `count` batches with ONE host synchronisation (device-drawn negatives): the
 * error passes of bin/train.py (train_error / validation_error, models.py:649-668, run before
 * training and after every epoch, train.py:262-348) are loops over all batches whose results
 * are only averaged.  losses_out[i] = the value sert_eval_batch would return for
 * batch_indices[i], in order; a non-finite value is reported by the caller after the call. */
int sert_eval_batches(sert_model* m, int split, const int64_t* batch_indices, int64_t count,
                      float* losses_out);

/* vectorspace predict_fn (models.py:1107-1118), batched over Q queries:
 * out[q] = tanh(avg[q] . W + b)   (no clip).  avg (Q, d_w), out (Q, d_e). */
int sert_predict_project(sert_model* m, const float* avg, int64_t num_queries, float* out);

/* loglinear predict_fn (models.py:880-890): ids (rows, n) -> per-token
 * distributions out (rows, n, V_e).  rows need not equal batch_size. */
int sert_predict_tokens(sert_model* m, const void* ids, int64_t rows, float* out);

/* ---- entity scoring (bin/query.py:239-370, batched) --------------------- */

typedef struct sert_scorer sert_scorer;

/* VectorSpaceCallback.__init__ (query.py:241-302): take the entity table
 * (V_e, d) f32 (host, un-normalised; not modified), L2-normalise a device copy
 * (query.py:270-274).  Replaces NearestNeighbors.fit (query.py:288-293). */
int sert_scorer_create(int device, const float* entities, int64_t num_entities, int32_t dim,
                       sert_scorer** out);
int sert_scorer_destroy(sert_scorer* s);

/* VectorSpaceCallback.process (query.py:320-367) for Q queries at once: L2-normalise
 * each projection (query.py:333-336), score every entity with (cos + 1)/2
 * (query.py:352-357) and return the k best per query, sorted by score
 * descending, ties by lowest entity index (replaces kneighbors / cdist+argsort,
 * query.py:304-318, and the Python candidate loop :348-365).
 *   proj (Q, d) f32 host;  idx_out (Q, k) int32;  score_out (Q, k) f32;  1 <= k <= min(V_e, 1024)
 * The ranking and the scores are those of the fp32 cosine.  For large tables (V_e >= 32768) the
 * candidates are found by a bf16 matrix-pipe pass whose proven error bound decides which
 * entities get the fp32 score; rows where that bound cannot separate the top k are computed in
 * fp32 throughout (csrc/kernels_score_bf16.h).  SERT_SCORE_FP32=1 disables the bf16 pass. */
int sert_scorer_topk(sert_scorer* s, const float* proj, int64_t num_queries, int32_t k,
                     int32_t* idx_out, float* score_out);

/* All scores (no selection): score_out (Q, V_e) f32 = (cos + 1)/2.  For --top
 * unset / > 1024 (query.py:250-260 ranks every entity); the caller orders them. */
int sert_scorer_scores(sert_scorer* s, const float* proj, int64_t num_queries, float* score_out);

/* Page-locked host memory for the arrays that cross this boundary on every query call (the
 * (Q, d) projections in, the (Q, k) indices and scores out).  The reference hands numpy arrays
 * to sklearn (query.py:304-318); a caller that builds its query block in such a buffer and reads
 * the results from one lets sert_scorer_topk move them at PCIe speed, asynchronously, instead of
 * through the driver's pageable staging copies (0.4 ms of a 1.25 ms C5 call). */
int sert_host_alloc(void** out, size_t bytes);
int sert_host_free(void* p);

/* Convenience: create + topk + destroy. */
int sert_score_topk(int device, const float* entities, int64_t num_entities, int32_t dim,
                    const float* proj, int64_t num_queries, int32_t k,
                    int32_t* idx_out, float* score_out);

/* ---- data parallel (new: the reference is single-device) ---- */

#define SERT_COMM_ID_BYTES 128
/* ncclGetUniqueId; rank 0 calls it and ships the bytes to the other ranks. */
int sert_comm_unique_id(char id[SERT_COMM_ID_BYTES]);
/* ncclCommInitRank on this handle's device.  From here on the model is data parallel
 * (SURVEY 8-e, 8-f4): rank r trains rows [r*B_l, (r+1)*B_l) of every global batch.  Every rank
 * OWNS 1/world of the word table (and of any other tensor beyond 4 M elements): the dense
 * optimiser (sert/models.py:548-549: every element, every step) runs on the owned share only and
 * only that share of the optimiser state exists.
 *   word table, default: owned BY ROWS; per step two all-to-alls over static per-batch lists --
 *     the parameter rows a rank's batch touches come from their owners before the forward, their
 *     gradient rows go back after the backward (csrc/kernels_xchg.h).  Rows nobody touches do not
 *     travel.  Until the next collective read, a rank's copy of R_w is current only where it owns
 *     or has fetched: sert_get_tensor(SERT_T_RW), the evaluation calls and sert_predict_tokens then
 *     first all-gather the table and are therefore COLLECTIVE (every rank, same order).
 *   SERT_DP_EXCHANGE=zero1 (or a word dim that is no multiple of 4, keep_grads, SERT_AR_CHUNKS > 1)
 *     and every other big tensor: gradient reduce-scattered, owned slabs updated, all-gathered.
 * The small tensors' gradients and the loss sum are all-reduced and those tensors updated
 * identically everywhere.  sert_get_tensor of a SERT_T_STATE* tensor is a COLLECTIVE call. */
int sert_comm_init(sert_model* m, const char id[SERT_COMM_ID_BYTES], int rank, int world);
/* Host-mediated exchange: every collective of the data-parallel step becomes device -> pinned host
 * -> fn -> device, synchronously.  fn is an ALL-TO-ALL of float segments: on entry `send` holds, for
 * every rank q (this rank included), send_counts[q] floats at send_offsets[q]; on return `recv` must
 * hold, at recv_offsets[q], the recv_counts[q] floats rank q addressed to this rank.  Offsets and
 * counts are in floats; fn moves bits and never does arithmetic on them.  Returns 0 on success.
 * Reduce-scatter, all-gather, all-reduce and the row exchange are all expressed through it, every
 * rank receiving exactly its pieces (sums are formed in rank order on the host).  A verification
 * transport, not a fast path: it lets several ranks share ONE GPU (RCCL refuses duplicate devices),
 * so the data-parallel step -- row sharding, global 1/B scaling, rank-invariant negatives, piece and
 * slab indexing of the owned tensors, L2 applied once, loss reduction -- can be checked against a
 * single-process run on a 1-GPU box. */
typedef int (*sert_alltoall_fn)(void* user, const float* send, const int64_t* send_offsets,
                                const int64_t* send_counts, float* recv, const int64_t* recv_offsets,
                                const int64_t* recv_counts);
int sert_comm_init_host(sert_model* m, int rank, int world, sert_alltoall_fn fn, void* user);
/* Exchange statistics of a data-parallel model, out[0..n) (n <= 8):
 *   [0] world  [1] exchange of the word table: 0 none, 1 ZeRO-1 (reduce-scatter + all-gather), 2 by rows
 *   [2] bytes this rank sent + received per training step (mean over the steps run so far)
 *   [3] the same figure ZeRO-1 would move: 2 x 2 (N-1)/N x padded table bytes
 *   [4] transport: 1 RCCL, 2 host-mediated  [5] training steps counted
 *   [6] mean rows per batch this rank fetches from peers  [7] mean rows per batch it serves */
int sert_comm_stats(sert_model* m, double* out, int n);
int sert_comm_destroy(sert_model* m);

/* ---- diagnostics -------------------------------------------------------- */

/* roctx ranges around host-side phases (no reference counterpart): forwarded to
 * roctxRangePushA / roctxRangePop of the ROCm tools library when it can be loaded, no-ops otherwise.
 * With SERT_ROCTX=1 in the environment the library itself wraps every kernel group of a step (the
 * sert_timing_name groups) in a range. */
int sert_profile_range_push(const char* name);
int sert_profile_range_pop(void);

/* hipStreamSynchronize on the handle's stream. */
int sert_synchronize(sert_model* m);
/* Average duration in microseconds of the named kernel group over the steps
 * since the last sert_timing_reset (HIP events on the handle's stream).
 * Enabled with sert_timing_enable(m, 1); while enabled the step runs fully serialised
 * on one stream (each kernel is measured alone) and every event record drains the
 * stream, so a timed step is slower than an untimed one: take throughput from
 * untimed steps.  Groups: sert_timing_count / sert_timing_name.
 * sert_timing_enable(m, 2): IN-STEP timing -- the normal schedule (all streams, run-ahead of the announced batch), every
 * plain kernel launch of a group bound to a (start, stop) event pair of its own (hipExtLaunchKernelGGL: the kernel's own
 * dispatch timestamps, no barrier packets); sert_timing_avg_us then returns the group's kernel time per training step as
 * it runs BESIDE the other queue's kernels, sert_timing_launches its timed launches per step.  Launches that carry a
 * completion event of the schedule are not timed (their groups read low or 0). */
int sert_timing_enable(sert_model* m, int on);
int sert_timing_reset(sert_model* m);
int sert_timing_count(sert_model* m);
const char* sert_timing_name(sert_model* m, int i);
double sert_timing_avg_us(sert_model* m, int i);
double sert_timing_launches(sert_model* m, int i);

#ifdef __cplusplus
}
#endif
#endif /* SERT_HIP_H */
