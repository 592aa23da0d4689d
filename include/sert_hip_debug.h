/*
 * sert_hip_debug.h -- test hooks and micro-benchmarks of libsert_hip.so.
 *
 * NOT part of the drop-in boundary (include/sert_hip.h is): nothing here replaces a reference function and no product code
 * path (sert_amd/models.py, inference.py, scoring.py, training.py, bin/) calls these.  They exist so that
 *   - every GEMM kernel of the library can be pinned against float64 on host arrays (tests/test_gpu_gemm.py),
 *   - the host-side index / exchange-list builders can be checked without a GPU (tests/test_word_index_cpu.py,
 *     tests/test_row_exchange_cpu.py),
 *   - bench.py can measure the denominators of its roofline fractions on the box it runs on (sert_bench_memory).
 * Same conventions as sert_hip.h (0 = ok, sert_last_error()).
 */
#ifndef SERT_HIP_DEBUG_H
#define SERT_HIP_DEBUG_H

#include "sert_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Launches of the word-table update (sert/models.py:548-549 applied to R_w) by kernel form since sert_create -- host
 * counters, test hook: out[0] dense (adam_l2 / adadelta_l2), out[1] dense_update_lazy, out[2..7] dense_update_skip
 * <32,1> <64,1> <32,3> <64,2> <64,3> <64,4> (lanes per row, float4 columns per lane), out[8] the dense_update_skip passes
 * that read and wrote every row, out[9] its sparse passes.  n <= 10.  The tests assert through this that every template
 * shape met the oracle (tests/test_gpu_skip_shapes.py). */
int sert_debug_update_counts(sert_model* m, int64_t* out, int n);

/* Test hook: overwrite the step's gradient scratch -- the flat buffer [g_Rw | g_Re | g_W | g_b | loss, sum of squares] with
 * quiet NaNs, the per-entity sorted-run bounds behind it with the wrong run [0, 1) -- after waiting for the device.  A step
 * whose negatives were drawn ahead launches NO prologue (nothing is zeroed): it relies on every value it reads having been
 * written by this step's own kernels.  A run with this call between the steps must equal the run without it bit for bit
 * (tests/test_gpu_parity.py::test_steps_read_nothing_stale_from_the_gradient_scratch).  Fails while a run-ahead step
 * (sert_hint_next_batch) is in flight: its gradients live there. */
int sert_debug_poison_scratch(sert_model* m);

/* Host-only (no device is touched): the row-exchange lists rank `rank` of `world` derives for batch
 * `batch` from the touched-row bitmaps of ALL ranks, allbits[world][num_batches][bit_words] -- the
 * function sert_upload_dataset runs on the gathered bitmaps (csrc/kernels_xchg.h).  Lets the
 * multi-rank algebra of the exchange be checked without any GPU: serve_cnt / fetch_cnt [world];
 * serve_rows / fetch_rows peer-major; union_rows with their contributions ptr / ent in rank order
 * (ent < 0: this rank's own row); sizes[5] = {serve, fetch, union, entries, largest transfer in rows}.
 * Every output array must hold `capacity` entries. */
int sert_debug_row_lists(const uint32_t* allbits, int world, int rank, int64_t num_batches, int64_t bit_words,
                         int64_t rows_per_rank, int64_t vocab, int64_t batch, int32_t* serve_cnt,
                         int32_t* fetch_cnt, int32_t* serve_rows, int32_t* fetch_rows, int32_t* union_rows,
                         int32_t* ptr, int32_t* ent, int64_t capacity, int64_t* sizes);

/* Host only, no GPU (test hook, no reference counterpart): the inverted index word -> batch rows that
 * sert_upload_dataset builds for a vectorspace model (the order-fixed replacement of Theano's AdvancedIncSubtensor1,
 * autodiff of sert/models.py:180), built for ids[num_batches][B][n] and EVALUATED ON THE HOST the way the segmented-sum
 * kernels walk it: grad_out[vocab][d] = the word-table gradient of batch `batch` for source rows src[B][d] (dh), i.e.
 * sum over the occurrences of a word of src[row] / divisor.  row_groups > 1: level 0 cut into row ranges (XCD lists);
 * dense_heavy: bit 0 = the batch's heaviest words summed outside the tree, bit 1 = level 0 sorted by item length with every
 * item's first row number in its descriptor (what the vectorspace models upload).  stats[8] = {levels, items, partial rows, final items,
 * dense words, row groups, level-0 items, distinct words}. */
int sert_debug_word_index_sum(const void* ids, int id_bytes, int64_t num_batches, int B, int n, int vocab, int row_groups,
                              int dense_heavy, int64_t batch, const float* src, int d, float divisor, float* grad_out,
                              int64_t* stats);

/* Micro-benchmark of the fp32 MFMA GEMM on device-resident random operands:
 * C (M,N) = op(A).op(B); ta/tb as in gemm.h; epi 0 = store, 1 = +bias, 2 = tanh(+bias);
 * splits > 1 = split-K partial slabs.  Returns the average launch time in *avg_us
 * (HIP events, `iters` launches after 2 warm-ups). */
int sert_bench_gemm(int device, int ta, int tb, int epi, int M, int N, int K, int splits,
                    int iters, double* avg_us);

/* The same dispatch on HOST arrays (test hook, no reference counterpart): C (M,N) = epi(op(A).op(B)), A (M,K) or
 * (K,M) if ta, B (K,N) or (N,K) if tb, bias (N) for epi 1 / 2 (only with ta = 0).  The shape is routed to the kernel a
 * training step would use for it, so every GEMM kernel of the library can be pinned against float64. */
int sert_debug_gemm(int device, int ta, int tb, int epi, int M, int N, int K, const float* A, const float* B,
                    const float* bias, float* C);

/* The split-K form of the same dispatch (test hook): out (M*N + N) = A^T.B, A (K,M), B (K,N) host arrays, followed by
 * the N column sums of B -- the split-K launch with the column sums riding along + the order-fixed combine that the
 * projection's dW / db take in a training step (sert/models.py:1057-1061, autodiff). */
int sert_debug_gemm_splitk(int device, int M, int N, int K, int splits, const float* A, const float* B, float* out);

/* ... and of a product A.op(B) over a long K cut into `splits` k ranges (test hook): C (M,N), A (M,K), B (K,N) or (N,K) if tb,
 * through the split launch + order-fixed combine of the loglinear dG = dZ.W^T over a large entity vocabulary
 * (sert/models.py:846-849, autodiff). */
int sert_debug_gemm_longk(int device, int tb, int M, int N, int K, int splits, const float* A, const float* B, float* C);

/* Memory-system micro-benchmarks: the denominators a step's memory-bound kernels are priced
 * against (no reference counterpart; measurement only).  Average launch time over `iters`
 * launches (HIP events on the launching stream, 2 warm-ups) in *avg_us.
 *   SERT_MEMBENCH_COPY       float4 stream copy: `bytes` read + `bytes` written per launch
 *   SERT_MEMBENCH_READ       float4 stream read of `bytes`
 *   SERT_MEMBENCH_GATHER     the step's own window gather (vs_gather_mean) over uniformly random
 *                            rows: `bytes` of output rows of `row_bytes`, each the mean of `window`
 *                            rows of a table of `table_bytes` -> bytes * window fetched per launch
 *   SERT_MEMBENCH_OPTIMIZER  the dense Adam kernel over four arrays of `bytes` (4 read, 3 written),
 *                            placed `gap_bytes` apart inside one allocation ((size_t)-1: four
 *                            allocations of their own, as a model holds them)
 * blocks: workgroups of the launch (0 = the kernel's default). */
enum { SERT_MEMBENCH_COPY = 0, SERT_MEMBENCH_READ = 1, SERT_MEMBENCH_GATHER = 2, SERT_MEMBENCH_OPTIMIZER = 3 };
int sert_bench_memory(int device, int kind, size_t bytes, size_t table_bytes, int row_bytes,
                      int window, size_t gap_bytes, int blocks, int iters, double* avg_us);

#ifdef __cplusplus
}
#endif
#endif /* SERT_HIP_DEBUG_H */
