#!/usr/bin/env python
"""bench.py -- training (word-window, entity) pairs/s of the SERT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one call of train_fn(batch_index) (sert/models.py:581-588): forward, backward,
L2, dense optimiser update on one batch of synthetic (window, entity) pairs, INCLUDING the
per-step loss read-back the reference's epoch loop performs (sert/models.py:369-379).

Workload at N = 1: BASELINE.json configs[1] -- the LSE config |V_w| = 100k, |V_e| = 1k, d = 128,
window = 10, batch = 65536 -- run with the reference's LSE model (VectorSpaceLanguageModel: NCE,
z = 10, Adam).  N > 1: one process per GPU (``python bench.py --gpus N`` starts them itself; a
launcher that sets RANK / LOCAL_RANK / WORLD_SIZE -- e.g. ``python -m torch.distributed.run`` --
works too; nothing here imports PyTorch), SURVEY 8-d / BASELINE configs[2]: the SAME configuration, global
batch fixed at 65536 and split by rows over the ranks ("scaling": "strong"; the word table is owned by rows and
the rows a batch touches travel in two all-to-alls per step, the small tensors in one all-reduce -- RCCL).  The
weak case (65536 rows PER GPU) is the `weak_scaling` sub-record.

Prints ONE JSON line (rank 0).  ``roofline`` describes the LONGEST kernel group of the step (HIP
events on the model's stream, measured live; ``traffic`` from two rocprofv3 --pmc passes this
script runs on itself), ``cpu_baseline`` the CPU restatements of the same step timed on this
node's host cores on a bounded sample: a multithreaded C implementation pinned to one socket
(the headline baseline) and the single-threaded numpy oracle.
"""
import argparse
import gc
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA DENSE peak (~2.5 PF; measured 2495)
COMMITTED_PMC = 'profiles/r03v_vs_c2_pmc.json'


def synth_data(rng, N, n, Vw, Ve):
    """Zipf(1.1) token ranks clipped to V_w then permuted once,
    labels ~ U[0,V_e), weights = 1."""
    ranks = np.minimum(rng.zipf(1.1, size=(N, n)) - 1, Vw - 1)
    perm = rng.permutation(Vw).astype(np.uint32)
    X = perm[ranks].astype(np.min_scalar_type(Vw - 1))
    y = rng.randint(0, Ve, size=N).astype(np.int32)
    w = np.ones(N, dtype=np.float32)
    return X, y, w


def glorot(rng, shape):
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


LAZY_MAX_TOUCHED, LAZY_K = 0.5, 4       # sert_hip.hip: SERT_LAZY_MAX behind an announced next batch; kernels_opt.h: kLazyK


def lazy_fractions(X, B, num_batches, Vw, dw):
    """The word-table update is LAZY where a batch touches <= 50 % of the rows and the next batch is announced (the timed
    loop announces every batch; kernels_opt.h: dense_update_skip): a row that neither the batch nor the announced next one
    touches is neither read nor written -- its share of sum(p^2) was left behind by the launch that wrote it -- except every
    4th update, which reads and writes every row.  Returns the fractions of rows touched / moved (read AND written) per
    step for the cyclic batch order of the timed loop -- what the launch really moves."""
    nb = min(num_batches, 8)          # (a sample of the batches: the fractions vary by < 1 % between batches of one stream)
    sets = [np.unique(X[j * B:(j + 1) * B]) for j in range(nb)]
    f_t = float(np.mean([len(u) for u in sets])) / Vw
    lazy_off = os.environ.get('SERT_LAZY_SKIP', '1') in ('0',)
    lazy_max = float(os.environ.get('SERT_LAZY_MAX', LAZY_MAX_TOUCHED))
    if f_t > lazy_max or Vw * dw < (1 << 22):
        # the dense launch (adam_l2 with the row filter): p, m, v read and written for every row, g read for touched rows only
        return {'lazy': False, 'touched': f_t}
    f_u = float(np.mean([len(np.union1d(sets[j], sets[(j + 1) % nb])) for j in range(nb)])) / Vw
    return {'lazy': True, 'touched': f_t, 'written': f_u + (1.0 - f_u) / LAZY_K, 'reads_every_row': lazy_off}


def x3_applies(M, N, K, ta=False):
    """gemm_x3.h: x3_shape_ok (the library's dispatch rule; SERT_GEMM_FP32=1 switches the bf16-pipe kernels off)."""
    if os.environ.get('SERT_GEMM_FP32', '0') not in ('', '0'):
        return False
    cdiv = lambda a, b: -(-a // b)
    if ta:
        tiles = cdiv(N, 160) if 128 < M <= 320 else cdiv(M, 128) * cdiv(N, 128)
        return M <= 4096 and N <= (1 << 20) and (K >= 4096 or (K >= 1024 and tiles >= 128))   # (the step's own launches are split: tiles x splits >= 16)
    tm = 128 if N <= 128 else 256
    tn = 128 if N <= 128 else (256 if (N <= 256 or (N > 320 and cdiv(N, 256) * 256 <= cdiv(N, 320) * 320)) else 320)
    big = K % 4 == 0 and (M >= 128 * tm or (K >= 256 and M >= 1024 and cdiv(M, tm) * cdiv(N, tn) >= 256))
    mid = K >= 256 and M >= 1024 and cdiv(M, 128) * cdiv(N, 128) >= (160 if (K % 4 == 0 and N % 4 == 0) else 96)
    return N <= (1 << 20) and K <= 4096 and (big or mid)


def group_work(kind, B, n, s, dw, de, Ve, Vw, z, distinct_words=None, shards=1, lazy=None):
    """Per timed kernel group: what it is priced against and how much work one launch does.

      kind 'mfma'   : algorithmic flops; the share gemm_x3.h runs as six bf16 products per fp32 product is priced on the
                      bf16 MFMA peak with 6 x the flops, the rest on the fp32 MFMA peak (frac = pipe time at peak / time)
      kind 'stream' : bytes streamed once (32 B per parameter and step for the optimiser)
                      -> HBM spec peak and the MEASURED achievable rate of the same access shape
      kind 'rows'   : data-dependent row fetches out of a table that fits the caches (L2 / Infinity
                      Cache): `fetch` = bytes of rows actually fetched per launch, priced against the
                      MEASURED row-fetch ceilings (bench.memory_ceilings) -- `alg` keeps SURVEY 8(d)'s
                      per-pair figure, which also counts the read-modify-write of gradient rows that
                      the order-fixed reductions keep in registers
    Keys = timing group names of libsert_hip.so (sert_timing_name).  Loglinear: the GEMMs and the
    gather run on the batch's DISTINCT words (duplicate tokens share a logit row), so their
    EXECUTED work is counted with U = distinct_words rows instead of B n."""
    def mfma(*gemms):
        # gemms: (M, N, K, ta) of every GEMM of the group; 2 M N K flops each, on the bf16 pipe where gemm_x3.h takes it
        fl = sum(2.0 * M * N * K for M, N, K, _ in gemms)
        fx = sum(2.0 * M * N * K for M, N, K, ta in gemms if x3_applies(M, N, K, ta))
        return dict(kind='mfma', flops=fl, flops_x3=fx)
    def stream(by, **kw): return dict(kind='stream', alg=by, **kw)
    def rows(alg, fetch, row_bytes, table_bytes, resident=None): return dict(
        kind='rows', alg=alg, fetch=fetch, row_bytes=row_bytes, table_bytes=table_bytes, resident=resident)
    # data parallel: a rank's dense update covers the word-table rows it owns (1/shards of them)
    P_w = 32.0 * Vw * dw / shards
    # lazy launch (single GPU): 12 B read + 12 B written per element of the rows that are moved (this batch's, the next
    # batch's, every row every 4th update), 4 B of gradient for the touched ones -- the bytes THIS launch's algorithm moves,
    # averaged over the cycle of four launches (the reference's dense update: 32 B)
    lazy_note = None
    if lazy is not None and shards == 1 and lazy.get('lazy'):
        # (dense_update_skip: 24 B per element of a row that is moved; SERT_LAZY_SKIP=0 -- dense_update_lazy -- reads all 12 B always)
        lazy_note = dict(lazy, reference_dense_bytes=P_w,
                         bytes_per_element=(12.0 + 12.0 * lazy['written'] if lazy.get('reads_every_row') else 24.0 * lazy['written']) + 4.0 * lazy['touched'])
        P_w = Vw * dw * lazy_note['bytes_per_element']
    elif lazy is not None and shards == 1:
        # dense launch: 12 B read + 12 B written per element, 4 B of gradient for the rows the batch touches (the others have
        # g = 0: neither zero-filled nor read) -- what THIS kernel's algorithm moves; SURVEY 8(d)'s 32 B is the reference's
        lazy_note = dict(lazy, reference_dense_bytes=P_w, bytes_per_element=24.0 + 4.0 * lazy['touched'])
        P_w = Vw * dw * lazy_note['bytes_per_element']
    if kind in ('vectorspace', 'vectorspace_softmax'):
        w = {
            'gather':               rows(B * (n * s + 4 * n * dw), B * 4.0 * n * dw, 4 * dw, 4.0 * Vw * dw),   # vs_gather_mean
            'gemm_fwd':             mfma((B, de, dw, False)),              # gemm_x3 / gemm_f32_mfma NN+tanh
            'gemm_dW':              mfma((dw, de, B, True)),               # gemm_x3 / gemm_f32_mfma TN split-K
            'splitk_combine':       stream(4.0 * 1024 * (dw * de + de)),   # reduce_partials
            'gemm_dX':              mfma((B, dw, de, False)),              # gemm_x3 / gemm_f32_mfma NT
            'gemm_bwd_fused':       dict(kind='mfma', flops=4.0 * B * dw * de, flops_x3=0.0),   # vs_bwd_fused: dh and dW in one launch (opt-in, fp32 MFMA)
            'word_grad_segsum':     rows(B * (8.0 * n * dw), B * 4.0 * n * dw, 4 * dw, 4.0 * B * dw),   # segsum_rows: rows of dh
            'optimizer_word_table': stream(P_w, optimizer_elems=Vw * dw // shards, **({'lazy': lazy_note} if lazy_note else {})),  # adam_l2 / dense_update_lazy (R_w)
            # adam_l2 over R_e where it is a big table (C4), else one optimizer_small launch: launch latency
            'optimizer_other':      (stream(32.0 * (Ve * de + dw * de + de), optimizer_elems=Ve * de) if Ve * de > (1 << 22)
                                     else dict(kind='latency', alg=32.0 * (Ve * de + dw * de + de))),
        }
        if kind == 'vectorspace':
            w.update({
                # vs_nce(_regs): (1+z) entity rows per pair out of R_e (+ the pair's own t row)
                'loss':               rows(B * (8 + 4.0 * (1 + z) * de), B * 4.0 * (2 + z) * de, 4 * de, 4.0 * Ve * de),
                'entity_sort':        dict(kind='latency', alg=B * (1 + z) * 16.0),            # egrad_bucket / csort_*
                # egrad_acc / egrad_chunk_reduce: one row of clip(t) per (pair, candidate); row groups of
                # <= 2 MB stay in one XCD's L2 on the sort-free path
                'entity_grad_reduce': rows(B * (8.0 * (1 + z) * de), B * 4.0 * (1 + z) * de, 4 * de, 4.0 * B * de,
                                           resident='l2' if Ve <= 2048 else None),
                'entity_grad_fixup':  stream(8.0 * Ve * de),                                   # egrad_fixup
            })
        else:   # full softmax over the entity vocabulary: logits / dR_e / dp GEMMs dominate
            w['gemm_fwd'] = mfma((B, de, dw, False), (B, Ve, de, False))
            w['entity_grad_reduce'] = mfma((Ve, de, B, True))
            w['gemm_dX'] = mfma((B, dw, de, False), (B, de, Ve, False))
            w['loss'] = stream(B * 8.0 * Ve)
        return w
    U = distinct_words if distinct_words else B * n
    return {
        'gather':               rows(U * (s + 4.0 * dw), U * 4.0 * dw, 4 * dw, 4.0 * Vw * dw),
        'gemm_fwd':             mfma((U, Ve, dw, False)),
        # ll_row_from_table: n rows of the (U, V_e) log-probability table per batch row, dJ written
        'loss':                 rows(B * (4.0 * n * Ve + 4.0 * Ve), B * 4.0 * n * Ve, 4 * Ve, 4.0 * U * Ve),
        'per_word_dz_sums':     rows(B * n * 4.0 * Ve + U * 4.0 * Ve, B * n * 4.0 * Ve, 4 * Ve, 4.0 * B * Ve),
        'gemm_dW':              mfma((dw, Ve, U, True)),
        'gemm_dX':              mfma((U, dw, Ve, False)),
        'word_grad_segsum':     rows(U * (8.0 * dw), U * 4.0 * dw, 4 * dw, 4.0 * U * dw),
        'optimizer_word_table': stream(P_w, optimizer_elems=Vw * dw // shards, **({'lazy': lazy_note} if lazy_note else {})),
        'optimizer_other':      (stream(32.0 * (dw * Ve + Ve), optimizer_elems=dw * Ve) if dw * Ve > (1 << 22)
                                 else dict(kind='latency', alg=32.0 * (dw * Ve + Ve))),
    }


# timing group -> HIP kernel(s) that dominate it, per model kind (names as rocprofv3 prints them;
# the first is the one a given configuration normally runs: e.g. the entity gradient is
# egrad_acc up to 2048 entities and egrad_chunk_reduce above)
_COMMON_KERNELS = {
    # (128x128 tiles; 64x64 tiles for fewer than / exactly two big tiles per CU; 128x160 tiles for d = 300)
    'gemm_dW': ('gemm_x3<true, false, 0', 'gemm_f32_mfma<true, false, 0', 'gemm_f32_mfma_n160<true, false, 0'), 'splitk_combine': ('reduce_partials',),
    'gemm_dX': ('gemm_x3<false, true, 0', 'gemm_f32_mfma<false, true, 0', 'gemm_f32_mfma_small<true, 0', 'gemm_f32_mfma_n160<false, true, 0'),
    'gemm_bwd_fused': ('vs_bwd_fused',), 'word_grad_segsum': ('segsum_rows<', 'segsum_rows_plus', 'segsum_upper_fused', 'segsum_heavy'),
    'optimizer_other': ('optimizer_small', 'adam_l2'), 'finalize': ('vs_tail', 'finalize_loss', 'sum_partial'),
}
# kernels a bench run launches that belong to no step: the memory micro-benchmarks (sert_bench_memory runs the step's own
# gather and dense Adam kernels on scratch arrays) and the runtime's fills / copies
NON_STEP_KERNELS = ('mb_', '__amd_rocclr', 'vs_gather_mean', 'adam_l2<false>')
KERNELS_OF_GROUP = {
    'vectorspace': dict(_COMMON_KERNELS, **{
        'gather': ('vs_gather_mean', 'vs_project_x3'),     # (vs_project_x3: a -DSERT_VARIANTS build with SERT_PROJ_FUSED=1 only)
        'prologue': ('vs_sample_negatives', 'sumsq_like_small'),      # (no timing group of its own: beside the forward)
        'gemm_fwd': ('gemm_x3<false, false, 2', 'gemm_f32_mfma<false, false, 2', 'gemm_f32_mfma_small<false, 2', 'gemm_f32_mfma_n160<false, false, 2'),
        'loss': ('vs_nce_regs', 'vs_nce'), 'entity_sort': ('egrad_bucket', 'csort_scatter', 'csort_'),
        'entity_grad_reduce': ('egrad_acc', 'egrad_chunk_reduce'),
        'entity_grad_fixup': ('egrad_group_sum', 'egrad_fixup'), 'optimizer_word_table': ('dense_update_skip', 'dense_update_lazy', 'adam_l2')}),
    'vectorspace_softmax': dict(_COMMON_KERNELS, **{
        'gather': ('vs_gather_mean',), 'gemm_fwd': ('gemm_x3<false, false, 2', 'gemm_x3<false, true, 0', 'gemm_f32_mfma<false, false, 0', 'gemm_f32_mfma<false, false, 2'),
        'loss': ('fs_softmax_ce',), 'entity_grad_reduce': ('gemm_f32_mfma<true, false, 0',),
        'optimizer_word_table': ('dense_update_skip', 'dense_update_lazy', 'adam_l2')}),
    'loglinear': dict(_COMMON_KERNELS, **{
        'gather': ('ll_gather_rows',), 'gemm_fwd': ('gemm_x3<false, false, 1', 'gemm_f32_mfma<false, false, 1',),
        'loss': ('ll_row_wave', 'll_row_from_table', 'll_fused_row', 'll_s_', 'll_logsoftmax_rows'),
        # (segsum_rows_plus_ll since round 5's commit 5ad725b: the heavy words' stream inside the tree's launches; the line of
        #  round 5 still named the kernel it replaced and its counter fields came out zero --
        #  tests/test_bench_line_cpu.py::test_every_profiled_kernel_maps_to_a_group keeps this map current)
        'per_word_dz_sums': ('segsum_rows_plus_ll', 'segsum_rows<64, true, true', 'segsum_rows_scalar<true', 'segsum_scalar_wave'),
        'optimizer_word_table': ('dense_update_skip', 'dense_update_lazy', 'adadelta_l2')}),
}


def kernels_of_group(kind, group):
    return KERNELS_OF_GROUP.get(kind, {}).get(group, ())


def group_of_kernel(kind, name):
    """The timing group a profiled kernel name belongs to ('' for the micro-benchmarks' and the runtime's kernels, None
    for a kernel the map does not know -- a renamed kernel: its group's counter fields would silently come out zero)."""
    best = None
    for group, prefixes in KERNELS_OF_GROUP.get(kind, {}).items():
        for pre in prefixes:
            if name.startswith(pre) and (best is None or len(pre) > best[0]):
                best = (len(pre), group)
    if best:
        return best[1]
    return '' if name.startswith(NON_STEP_KERNELS) else None


def build_model(kind, models, B_global, n, Vw, Ve, dw, de, z, X, y, w, seed):
    np.random.seed(seed)
    rng = np.random.RandomState(seed + 1)
    Rw = glorot(rng, (Vw, dw))
    empty_x = np.zeros((0, n), dtype=X.dtype)
    empty_y = np.zeros((0,), dtype=np.int32)
    if kind == 'vectorspace':
        Re = glorot(rng, (Ve, de))
        m = models.VectorSpaceLanguageModel(
            batch_size=B_global, window_size=n, num_negative_samples=z,
            representations_init=Rw, entity_representations_init=Re,
            regularization_lambda=0.01, training_set=(X, y, w),
            validation_set=(empty_x, empty_y))
    elif kind == 'vectorspace_softmax':
        Re = glorot(rng, (Ve, de))
        m = models.VectorSpaceSoftmaxLanguageModel(
            batch_size=B_global, window_size=n, representations_init=Rw,
            entity_representations_init=Re, regularization_lambda=0.01,
            training_set=(X, y, w), validation_set=(empty_x, empty_y))
    else:
        m = models.LanguageModel(
            batch_size=B_global, window_size=n, representations_init=Rw,
            output_layer_size=Ve, regularization_lambda=0.01,
            training_set=(X, y, w), validation_set=(empty_x, empty_y))
    return m


CLOCK_STEPS = 100   # untimed steps in front of a throughput pass' own warm-up: ~30 ms of full load (see timed_steps)


def timed_steps(model, dist, num_batches, steps, warmup, timing=True):
    eng = model._engine
    if not timing:
        # The event-bracketed pass (one queue, every kernel alone) and the micro-benchmarks in front of a throughput pass
        # leave the GPU below its operating clocks, and W = 5 warm-up steps are 1.4 ms: one 20-step line of this round
        # read 0.306 ms where every other pass of the same process read 0.264-0.268.  So the model
        # first runs CLOCK_STEPS untimed steps; then the W warm-up steps and exactly K timed steps of the contract.
        # (hinted like the timed steps: without the announcement every lazy word-table update writes every row, and the
        #  counter passes -- which average over ALL launches of a kernel -- read 1.41 x the bytes the timed mode moves:
        #  round 5, tools/experiments/r05_lazy_writes.sh)
        for i in range(CLOCK_STEPS):
            eng.hint_next_batch((i + 1) % num_batches)
            model.train_fn(i % num_batches)
    for i in range(warmup):
        eng.hint_next_batch((i + 1) % num_batches if i + 1 < warmup else warmup % num_batches)
        model.train_fn(i % num_batches)
    if timing:
        # two instrumented steps before the instrumented ones that count: the first one through the serial
        # (one queue, event-bracketed) path pays one-time costs inside its timing groups
        eng.timing_enable(True)
        for i in range(2):
            eng.hint_next_batch((warmup + i + 1) % num_batches)
            model.train_fn((warmup + i) % num_batches)
    eng.timing_reset()
    eng.timing_enable(timing)
    eng.synchronize()
    dist.barrier()
    gc_was = gc.isenabled()
    gc.disable()        # (no collector pause inside a 5 ms timed region)
    t0 = time.perf_counter()
    last = 0.0
    for i in range(steps):
        # as ModelInterface._iterate_batches does: announce the batch that follows
        eng.hint_next_batch((warmup + i + 1) % num_batches if i + 1 < steps else None)
        last = model.train_fn((warmup + i) % num_batches)
    eng.hint_next_batch(None)
    eng.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    eng.timing_enable(False)
    if not np.isfinite(last):
        raise RuntimeError('non-finite loss in the timed region')
    timings = eng.timings()
    if model._engine.cfg.kind == 1 and timings.get('gemm_dX', 0) > 0 and not timings.get('gemm_dW', 0) > 0:
        # SERT_BWD_FUSED=1: dh and dW came out of one launch (vs_bwd_fused), timed under gemm_dX
        timings['gemm_bwd_fused'] = timings.pop('gemm_dX')
    if model._engine.cfg.kind == 0 and 'entity_grad_reduce' in timings:
        # loglinear reuses this timing slot for the per-distinct-word sums of dJ / r
        timings['per_word_dz_sums'] = timings.pop('entity_grad_reduce')
    return dist.all_reduce_max(dt), timings, float(last)


def instep_pass(model, num_batches, steps):
    """Kernel time per group AS IT RUNS IN THE STEP: the normal schedule (both queues, run-ahead of the announced batch,
    per-step loss read-back) with every plain kernel launch bound to a (start, stop) HIP event pair of its own
    (sert_timing_enable(m, 2): hipExtLaunchKernel, the dispatch's own timestamps, no barrier packets).  The pass
    `timed_steps(timing=True)` measures every group ALONE on one queue; beside the side stream's entity chain the
    HBM-bound word-table update takes longer than alone (round 5: 55 against 47 us), and that is the duration the
    roofline fraction of the line is taken on.  Returns ({group: us per step}, {group: timed launches per step})."""
    eng = model._engine
    for i in range(4):
        eng.hint_next_batch((i + 1) % num_batches)
        model.train_fn(i % num_batches)
    eng.timing_reset()
    eng.timing_enable(2)
    try:
        for i in range(steps):
            eng.hint_next_batch((4 + i + 1) % num_batches if i + 1 < steps else None)
            model.train_fn((4 + i) % num_batches)
        eng.hint_next_batch(None)
        eng.synchronize()
        us, launches = eng.timings(), eng.timing_launches()
    finally:
        eng.timing_enable(0)
        eng.timing_reset()
    if eng.cfg.kind == 0 and 'entity_grad_reduce' in us:
        us['per_word_dz_sums'] = us.pop('entity_grad_reduce')
        launches['per_word_dz_sums'] = launches.pop('entity_grad_reduce')
    return ({k: v for k, v in us.items() if v > 0}, {k: v for k, v in launches.items() if v > 0})


_CEIL_CACHE = {}


def memory_ceilings(_capi, row_bytes, table_bytes_list, optimizer_elems=(), device=0):
    """What this box's memory system delivers (sert_bench_memory, HIP events, run in this process):
    float4 stream copy and read (achievable HBM), the step's own window gather over uniformly random
    rows of `row_bytes` out of tables of the given sizes (the L2-resident one is the row-fetch PEAK, the
    others the Infinity-Cache / HBM rates a table of that size sustains), and the dense optimiser's
    seven streams over tensors of the given element counts."""
    MB = 1 << 20
    out = _CEIL_CACHE.setdefault('stream', {})
    if not out:
        us = _capi.bench_memory(_capi.MEMBENCH_COPY, 1200 * MB, blocks=4096, iters=10, device=device)
        out['copy_GBps'] = 2 * 1200 * MB / (us * 1e-6) / 1e9
        us = _capi.bench_memory(_capi.MEMBENCH_READ, 1200 * MB, blocks=4096, iters=10, device=device)
        out['read_GBps'] = 1200 * MB / (us * 1e-6) / 1e9
    res = dict(out)
    rows = _CEIL_CACHE.setdefault('rows', {})
    want = [(row_bytes, 512 * 1024)] + [(row_bytes, int(t)) for t in table_bytes_list]
    for rb, tb in want:
        rb16 = max(16, (int(rb) + 15) // 16 * 16)
        key = (rb16, max(tb, rb16))
        if key not in rows:
            out_bytes = 65536 * rb16
            us = _capi.bench_memory(_capi.MEMBENCH_GATHER, out_bytes, table_bytes=key[1], row_bytes=rb16, window=10, iters=10, device=device)
            rows[key] = out_bytes * 10 / (us * 1e-6) / 1e9
    res['row_fetch_GBps'] = {'%dB_rows_of_%.1fMB' % (k[0], k[1] / 1e6): round(v, 1) for k, v in rows.items() if k[0] == max(16, (int(row_bytes) + 15) // 16 * 16)}
    opt = _CEIL_CACHE.setdefault('opt', {})
    for n in optimizer_elems:
        n = int(n)
        if n >= (1 << 16) and n not in opt:
            us = _capi.bench_memory(_capi.MEMBENCH_OPTIMIZER, n * 4, gap_bytes=_capi.SEPARATE_ALLOCATIONS,
                                    blocks=4096 if n >= (1 << 24) else 2048, iters=10, device=device)
            opt[n] = 28.0 * n / (us * 1e-6) / 1e9
    res['optimizer_stream_GBps'] = {str(n): round(v, 1) for n, v in opt.items() if n in [int(x) for x in optimizer_elems]}
    return res


def ceilings_for(_capi, work, device=0):
    """memory_ceilings for every row width / table size / optimiser tensor a work table names."""
    merged = {}
    opt = [w_.get('optimizer_elems', 0) for w_ in work.values() if w_['kind'] == 'stream']
    widths = sorted(set(w_['row_bytes'] for w_ in work.values() if w_['kind'] == 'rows')) or [512]
    for rb in widths:
        c = memory_ceilings(_capi, rb, [w_['table_bytes'] for w_ in work.values() if w_['kind'] == 'rows' and w_['row_bytes'] == rb], opt, device=device)
        rf = dict(merged.get('row_fetch_GBps', {}))
        rf.update(c['row_fetch_GBps'])
        merged.update(c)
        merged['row_fetch_GBps'] = rf
    return merged


def _row_ceiling(row_bytes, table_bytes):
    rows = _CEIL_CACHE.get('rows', {})
    rb16 = max(16, (int(row_bytes) + 15) // 16 * 16)
    l2 = rows.get((rb16, max(512 * 1024, rb16)))
    same = [(abs(k[1] - table_bytes), v) for k, v in rows.items() if k[0] == rb16]
    return l2, (min(same)[1] if same else None)


def kernel_table(timings, work, traffic=None):
    """Per timing group: HIP-event average, work per launch, achieved rate and its fraction of the
    bound it is priced against.  No fraction is taken against a peak the kernel does not run on:

      mfma   : executed flops / time / fp32 MFMA dense peak
      hbm    : streamed bytes / time / 8 TB/s spec, plus frac_of_achievable against the MEASURED rate of
               the same access shape on this box (stream copy, or the optimiser's seven streams over a
               tensor of the same size) -- with the PMC-counted bytes as numerator where a counter
               pass ran.  A streaming group whose counters show < 0.7 of its algorithmic bytes crossing
               the fabric is not HBM-bound: relabelled 'cache'
      cache  : row-fetch bytes / time / the MEASURED L2 row-fetch peak (uniformly random rows of the same
               width out of an L2-resident table, the gather kernel's own access shape); the rate a
               table of the kernel's real source size sustains under uniformly random ids is given
               beside it (`table_ceiling_GBps`; a Zipfian id stream can beat that one, never the peak)
    """
    kernels = {}
    traffic = traffic or {}
    stream_ceil = _CEIL_CACHE.get('stream', {})
    for name, us in timings.items():
        if us <= 0:
            continue
        wk = work.get(name)
        counted = traffic.get(name, {}).get('hbm_bytes')
        if wk is None or wk['kind'] == 'latency':
            kernels[name] = dict(us=round(us, 2), bound='latency')
            if counted is not None:
                kernels[name]['hbm_bytes_pmc'] = counted
            continue
        t = us * 1e-6
        if wk['kind'] == 'mfma':
            ach = wk['flops'] / t / 1e12
            fx = wk.get('flops_x3', 0.0)
            # time the launch's MFMAs occupy the matrix pipe at peak: the fp32 products that run as six bf16 products
            # (gemm_x3.h) at the bf16 rate, the others at the fp32 MFMA rate
            t_pipe = (wk['flops'] - fx) / (MFMA_F32_PEAK_TFLOPS * 1e12) + 6.0 * fx / (MFMA_BF16_PEAK_TFLOPS * 1e12)
            kernels[name] = dict(us=round(us, 2), bound='mfma', achieved=round(ach, 2), unit='TFLOP/s',
                                 frac=round(t_pipe / t, 4), algorithmic_flops=wk['flops'])
            if fx > 0:
                kernels[name].update(
                    pipe='bf16 MFMA, three exact bf16 pieces per fp32 operand, six products per fp32 product (gemm_x3.h)',
                    executed_bf16_tflops=round(6.0 * fx / t / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS,
                    share_on_bf16_pipe=round(fx / wk['flops'], 3),
                    frac_is='matrix-pipe time at peak (6 x the flops at the bf16 dense peak; any fp32-MFMA share at 157.3 TF) / launch time',
                    times_the_fp32_mfma_peak=round(ach / MFMA_F32_PEAK_TFLOPS, 3))
            else:
                kernels[name]['peak'] = MFMA_F32_PEAK_TFLOPS
            if counted is not None:
                kernels[name]['hbm_bytes_pmc'] = counted
            continue
        kind = wk['kind']
        if kind == 'stream' and counted is not None and counted < 0.7 * wk['alg']:
            kind = 'rows'     # (served by the caches: price the bytes it asks for as row fetches)
            wk = dict(wk, fetch=wk['alg'], row_bytes=512, table_bytes=0, resident=None)
        if kind == 'stream' and wk['alg'] / t / 1e9 > HBM_PEAK_GBS:
            # a "stream" that outruns the HBM peak is not coming from HBM (a small tensor that lives in L2 / the
            # Infinity Cache from one step to the next): no fraction of the HBM peak is taken
            kernels[name] = dict(us=round(us, 2), bound='cache', achieved=round(wk['alg'] / t / 1e9, 1), unit='GB/s', frac=None,
                                 algorithmic_bytes=wk['alg'],
                                 served_from='L2 / Infinity Cache (the algorithmic rate exceeds the HBM peak: the tensor is cache-resident)')
            if counted is not None:
                kernels[name]['hbm_bytes_pmc'] = counted
            continue
        if kind == 'stream':
            ach = wk['alg'] / t / 1e9
            rec = dict(us=round(us, 2), bound='hbm', achieved=round(ach, 1), unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4),
                       algorithmic_bytes=wk['alg'])
            if wk.get('lazy') and not wk['lazy'].get('lazy'):
                rec['dense_update'] = dict(wk['lazy'], note='algorithmic_bytes = 24 B per element + 4 B of gradient per element of a touched '
                                           'row (what adam_l2 with the row filter moves); reference_dense_bytes = the 32 B per parameter '
                                           'of SURVEY 8(d) (zero-fill + dense g read included)')
            elif wk.get('lazy'):
                rec['lazy_update'] = dict(wk['lazy'], note='lazy dense update (dense_update_skip): rows neither this nor the next batch touches are '
                                          'neither read nor written -- their share of sum(p^2) was left behind when they were written -- '
                                          'except every 4th update, which moves every row; algorithmic_bytes = what a launch moves on '
                                          'average over that cycle, reference_dense_bytes = 32 B per parameter of the reference\'s dense update')
            # achievable = the best streaming rate measured on this box for this shape: the read-only stream
            # bounds every read/write mix from above; the optimiser's own seven streams over a tensor of the
            # same size can beat it where the Infinity Cache holds part of the tensor
            opt_rate = _CEIL_CACHE.get('opt', {}).get(int(wk.get('optimizer_elems') or 0))
            achievable = max([x for x in (opt_rate, stream_ceil.get('read_GBps'), stream_ceil.get('copy_GBps')) if x] or [0])
            if achievable:
                moved = counted if counted is not None else wk.get('executed', wk['alg'])
                own = min(moved, wk['alg']) / t / 1e9
                rec['achievable_is'] = ('adam_l2 over a tensor of the same size' if opt_rate and opt_rate >= achievable
                                        else 'float4 read stream over 1.2 GB')
                if own > achievable:
                    # the kernel outran every stream measured beside it: the box's ceiling is at least its own
                    # rate, and that is what the fraction is taken against
                    rec['achievable_is'] = 'this kernel (the %s reached %.1f GB/s)' % (rec['achievable_is'], achievable)
                    achievable = own
                rec['achievable_GBps'] = round(achievable, 1)
                rec['frac_of_achievable'] = round(own / achievable, 4)
            kernels[name] = rec
        else:
            ach = wk['fetch'] / t / 1e9
            l2, same = _row_ceiling(wk['row_bytes'], wk['table_bytes'])
            rec = dict(us=round(us, 2), bound='cache', achieved=round(ach, 1), unit='GB/s', row_fetch_bytes=wk['fetch'],
                       algorithmic_bytes=wk['alg'])
            if l2:
                rec['peak_is'] = 'measured L2 row-fetch rate (uniformly random %d-byte rows of an L2-resident table)' % wk['row_bytes']
                if ach > l2:
                    # the kernel outran the ceiling kernel measured beside it: the ceiling of this box is at least
                    # the kernel's own rate, and that is what the fraction is taken against (never above 1)
                    rec['peak_is'] = 'this kernel (the %s reached %.1f GB/s)' % (rec['peak_is'], l2)
                    l2 = ach
                rec['peak'] = round(l2, 1)
                rec['frac'] = round(ach / l2, 4)
            if same and not wk.get('resident'):
                rec['table_ceiling_GBps'] = round(same, 1)
            kernels[name] = rec
        if counted is not None:
            kernels[name]['hbm_bytes_pmc'] = counted
    return kernels


LAUNCHES_OF_CHAIN = {'word_grad_segsum': 2, 'entity_sort': 3}   # (level 0 + the dense heavy words' stream; level 1 + their combine)


def roofline_of(kernels, traffic_by_group=None, traffic_source=None, kind='vectorspace', instep=None):
    """The kernel with the longest average launch (whatever it is).  frac = achieved / peak with achieved =
    algorithmic work / measured time (the contract's definition); frac_counter = the same with
    the PMC-counted bytes (what the memory system really moved); achievable_peak / frac_of_achievable:
    against the rate this box's memory system was measured to deliver for the same access shape."""
    cand = [k for k in kernels if kernels[k].get('bound') in ('hbm', 'mfma', 'cache')]
    # by average LAUNCH duration: the groups that are chains of launches of one kernel name (the three levels of the word
    # gradient's tree, the three kernels x digits of the counting sort) count with their time per launch
    dom = max(cand, key=lambda k: kernels[k]['us'] / LAUNCHES_OF_CHAIN.get(k, 1))
    kd = kernels[dom]
    mem = kd['bound'] != 'mfma'
    names = kernels_of_group(kind, dom)
    if kd['bound'] == 'cache':
        out = dict(kernel=dom, hip_kernel=names[0] if names else None, bound='hbm',
                   achieved=kd['achieved'], peak=kd.get('peak'), unit='GB/s', frac=kd.get('frac'), avg_us=kd['us'], traffic=None,
                   served_from='L2 / Infinity Cache: priced against the measured L2 row-fetch peak, not the HBM peak',
                   table_ceiling_GBps=kd.get('table_ceiling_GBps'))
    else:
        out = dict(kernel=dom, hip_kernel=names[0] if names else None, bound='hbm' if mem else 'mfma',
                   achieved=kd['achieved'], peak=HBM_PEAK_GBS if mem else kd.get('peak', MFMA_F32_PEAK_TFLOPS),
                   unit=kd['unit'], frac=kd['frac'], avg_us=kd['us'], traffic=None)
        if 'achievable_GBps' in kd:
            out['achievable_peak'] = kd['achievable_GBps']
            out['achievable_is'] = kd.get('achievable_is')
            out['frac_of_achievable'] = kd['frac_of_achievable']
    tr = (traffic_by_group or {}).get(dom)
    if tr:
        if tr.get('hip_kernel'):
            out['hip_kernel'] = tr['hip_kernel']
        out['traffic'] = tr['hbm_bytes']
        out['traffic_source'] = traffic_source
        out['avg_us_profiled'] = tr['avg_us_profiled']
        if mem and kd.get('algorithmic_bytes'):
            out['frac_counter'] = round(tr['hbm_bytes'] / (kd['us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            out['traffic_over_algorithmic'] = round(tr['hbm_bytes'] / kd['algorithmic_bytes'], 3)
    # the IN-STEP duration (instep_pass: the launch beside the other queue's kernels, HIP events of its own dispatch) is the
    # one `frac`, `achieved` and `frac_counter` are taken on; the kernel-alone figures stay beside them (round-5 verdict)
    us_in = (instep or {}).get(dom)
    if us_in and out.get('frac') is not None and out.get('achieved') is not None:
        scale = kd['us'] / us_in
        out['avg_us_alone'], out['frac_alone'], out['achieved_alone'] = out['avg_us'], out['frac'], out['achieved']
        out['avg_us'] = round(us_in, 2)
        out['frac'] = round(out['frac_alone'] * scale, 4)
        out['achieved'] = round(out['achieved_alone'] * scale, 1)
        if out.get('frac_counter') is not None:
            out['frac_counter_alone'] = out['frac_counter']
            out['frac_counter'] = round(out['frac_counter_alone'] * scale, 4)
        out['frac_is'] = ('on avg_us = the launch IN THE STEP, beside the other queue (HIP events of its own dispatch over K '
                          'steps of the normal schedule); *_alone = the same launch alone on the device')
    elif instep is not None:
        out['frac_is'] = 'on avg_us = the kernel ALONE on the device (no in-step sample for this group)'
    return out


# ---- live PMC passes --------------------------------------------------------------------
def pmc_traffic_live(inner_args, timeout=300, steps=13):
    """HBM bytes per launch of every kernel of a step: two rocprofv3 passes (FETCH_SIZE and
    WRITE_SIZE do not fit one TCC pass; --kernel-trace only beside them, as
    MI355X_MICROARCH.md prescribes) over a short run of THIS script's inner loop on the workload
    `inner_args` describes (--model / --batch / --vocab / ...), read back with tools/rocpd_pmc.py's
    corrections (FETCH_SIZE x 2 on gfx950)."""
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import rocpd_pmc
    tmp = tempfile.mkdtemp(prefix='sert_pmc_')
    try:
        dbs = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-d', out, '-o', 'p', '--',
                   sys.executable, os.path.abspath(__file__), '--profile-inner', '--steps', str(steps), '--warmup', '3'] + \
                  [str(a) for a in inner_args]
            env = dict(os.environ, TMPDIR='/tmp')
            for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('.db')]
            if r.returncode != 0 or not found:
                return None, 'rocprofv3 --pmc %s failed (rc %d): %s' % (counter, r.returncode, r.stderr.decode()[-300:])
            dbs[counter] = found[0]
        fetch = rocpd_pmc.per_kernel(dbs['FETCH_SIZE'], 'FETCH_SIZE')
        write = rocpd_pmc.per_kernel(dbs['WRITE_SIZE'], 'WRITE_SIZE')
        per = {}
        for k in set(fetch) | set(write):
            f = fetch.get(k, {}).get('kib', 0.0) * 1024
            w = write.get(k, {}).get('kib', 0.0) * 1024
            per[k] = dict(hbm_bytes=2 * f + w, fetch_bytes_corrected=2 * f, write_bytes=w,
                          avg_us_profiled=fetch.get(k, write.get(k, {})).get('us', 0.0),
                          calls=fetch.get(k, write.get(k, {})).get('calls', 0))
        return per, 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench (%d steps each)' % steps
    except Exception as e:   # noqa: BLE001 -- profiling is best effort, the bench line must still appear
        return None, 'live PMC passes failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def traffic_by_group(per_kernel, timings, kind='vectorspace'):
    """Match the profiled kernels ('name @grid') to the timing groups: the launch of the group's
    dominant kernel whose profiled duration is closest to the live HIP-event average."""
    out = {}
    if per_kernel:
        # dense_update_skip alternates between launches that move every row and launches that move what two batches touch
        # (kernels_opt.h); rocpd_pmc splits launches of one kernel and grid by duration ('#0', '#1'): merged back here,
        # weighted by calls -- the live average it is compared with is over the same mix
        merged = {}
        for key, rec in per_kernel.items():
            base = key.split(' #')[0]
            if key.startswith('dense_update_skip') and base != key:
                merged.setdefault(base, []).append(rec)
        for base, recs in merged.items():
            calls = float(sum(r.get('calls', 0) for r in recs)) or 1.0
            per_kernel = {k: v for k, v in per_kernel.items() if k.split(' #')[0] != base}
            per_kernel[base] = dict(calls=int(calls), merged_duration_clusters=len(recs),
                                    **{f: sum(r.get(f, 0.0) * r.get('calls', 0) for r in recs) / calls
                                       for f in ('hbm_bytes', 'fetch_bytes_corrected', 'write_bytes', 'avg_us_profiled')})
    for group, us in timings.items():
        kd = {'us': us}
        prefixes = kernels_of_group(kind, group)
        if not prefixes or not per_kernel:
            continue
        best = None
        for prefix in prefixes:
            for key, rec in per_kernel.items():
                if key.startswith(prefix):
                    d = abs(rec.get('avg_us_profiled', 0.0) - kd['us'])
                    if best is None or d < best[0]:
                        best = (d, dict(rec, hip_kernel=key.split(' @')[0]))
            if best:
                break   # the configuration's own kernel was profiled: do not fall through to the alternative
        if best:
            out[group] = best[1]
    return out


# ---- CPU baselines ----------------------------------------------------------------------
def cpu_baseline_port(B, n, Vw, Ve, dw, de, z, budget_s):
    """The numpy oracle (single-threaded restatement of the reference graph)."""
    from oracle import sert_oracle as O
    rng = np.random.RandomState(123)
    X, y, w = synth_data(rng, B, n, Vw, Ve)
    ora = O.VectorSpaceOracle(B, n, z, glorot(rng, (Vw, dw)), glorot(rng, (Ve, de)), glorot(rng, (dw, de)),
                              np.zeros(de, np.float32), 0.01)
    t0 = time.perf_counter()
    steps = 0
    while True:
        ora.train_step(X, y, w, rng.randint(0, Ve, size=(B, z)))
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 8:
            break
    return dict(value=steps * B / dt, unit='pairs/s', cores=1, kind='port',
                sample='%d steps of B=%d (%.1f s): numpy restatement of the reference graph (oracle/sert_oracle.py), '
                       'effectively single-threaded (fancy-index gather, ufunc.at scatter-add and the elementwise '
                       'optimiser are serial numpy kernels; the BLAS-threaded matmuls are < 2 %% of a step)' % (steps, B, dt))


CPU_MT_WORKER = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import cpu_baseline as CB
cores = CB.one_socket_cores()
os.sched_setaffinity(0, cores)
import bench
B, n, Vw, Ve, dw, de, z, budget = %(B)d, %(n)d, %(Vw)d, %(Ve)d, %(dw)d, %(de)d, %(z)d, %(budget)f
rng = np.random.RandomState(123)
nb = 4
X, y, w = bench.synth_data(rng, nb * B, n, Vw, Ve)
try:
    cpu = CB.VectorSpaceCPU(B, n, z, bench.glorot(rng, (Vw, dw)), bench.glorot(rng, (Ve, de)), bench.glorot(rng, (dw, de)),
                            np.zeros(de, np.float32), 0.01, native=True, out_dir=%(tmp)r)
    build = 'gcc -O3 -march=native -fopenmp (built on this node)'
except Exception:
    cpu = CB.VectorSpaceCPU(B, n, z, bench.glorot(rng, (Vw, dw)), bench.glorot(rng, (Ve, de)), bench.glorot(rng, (dw, de)),
                            np.zeros(de, np.float32), 0.01)
    build = 'gcc -O3 -march=x86-64-v3 -fopenmp (prebuilt)'
Xi = X.astype(np.int32)
for s in range(nb):
    cpu.index_batch(s, Xi[s * B:(s + 1) * B])          # static slices: indexed once, as the GPU engine does at upload
negs = [rng.randint(0, Ve, size=(B, z)).astype(np.int32) for _ in range(4)]
for s in range(2):
    cpu.train_step(Xi[s * B:(s + 1) * B], y[s * B:(s + 1) * B], w[s * B:(s + 1) * B], negs[s], slot=s)
t0 = time.perf_counter(); steps = 0; loss = 0.0; best = None
while True:
    s = steps %% nb
    loss = cpu.train_step(Xi[s * B:(s + 1) * B], y[s * B:(s + 1) * B], w[s * B:(s + 1) * B], negs[steps %% 4], slot=s)
    steps += 1
    ph = cpu.phases_ms()
    if best is None or ph['total'] < best['total']:
        best = ph
    dt = time.perf_counter() - t0
    if dt > budget or steps >= 400:
        break
assert np.isfinite(loss)
print('RESULT ' + json.dumps(dict(value=steps * B / dt, steps=steps, seconds=dt, cores=len(cores), threads=cpu.threads,
                                  build=build, ms_per_step=1000 * dt / steps, phases_ms=best)))
'''


def _cgroup_quota():
    """CPUs the container may use according to its cgroup (cpu.max), or None."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()[:2]
        return None if q == 'max' else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline_mt(B, n, Vw, Ve, dw, de, z, budget_s):
    """oracle/sert_cpu.c (OpenMP) in its own process, pinned to the physical cores of socket 0."""
    tmp = tempfile.mkdtemp(prefix='sert_cpu_')
    try:
        code = CPU_MT_WORKER % dict(root=ROOT, B=B, n=n, Vw=Vw, Ve=Ve, dw=dw, de=de, z=z, budget=budget_s, tmp=tmp)
        from oracle import cpu_baseline as CB
        cores = CB.one_socket_cores()
        # one OpenMP thread per physical core of socket 0, each bound to its own CPU (explicit
        # list: with OMP_PLACES=cores under a restricted affinity mask libgomp piled threads up)
        env = dict(os.environ, OMP_NUM_THREADS=str(len(cores)), GOMP_CPU_AFFINITY=' '.join(str(c) for c in cores),
                   OMP_DYNAMIC='false')
        env.pop('OMP_PLACES', None)
        env.pop('OMP_PROC_BIND', None)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
        r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT,
                           env=env, timeout=max(120.0, 6 * budget_s))
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith('RESULT ')]
        if r.returncode != 0 or not lines:
            return dict(value=None, kind='port', error=r.stderr.decode()[-400:])
        res = json.loads(lines[-1][len('RESULT '):])
        return dict(value=res['value'], unit='pairs/s', cores=res['cores'], kind='port',
                    ms_per_step=res['ms_per_step'], phases_ms_fastest_step=res.get('phases_ms'),
                    sample='%d steps of B=%d (%.1f s) of the same workload: multithreaded C restatement of the reference '
                           'graph (oracle/sert_cpu.c, %s; OpenMP, %d threads on the %d CPUs this process may use -- the '
                           'container\'s cgroup quota (cpu.max) on a host that shows %d hardware threads, not a whole '
                           'socket; the fastest unthrottled 64-thread step measured on such a host was 13.5 ms = 4.9 M '
                           'pairs/s, DESIGN.md section 4; order-fixed segmented sums, fused L2 + Adam) -- not Theano, '
                           'which cannot run here'
                           % (res['steps'], B, res['seconds'], res['build'], res['threads'], res['cores'], os.cpu_count() or 0),
                    cgroup_cpu_quota=_cgroup_quota(), unthrottled_64_thread_pairs_per_s_measured_earlier=4.9e6)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(B, n, Vw, Ve, dw, de, z, budget_s):
    host = len(os.sched_getaffinity(0))
    mt = cpu_baseline_mt(B, n, Vw, Ve, dw, de, z, budget_s)
    port = cpu_baseline_port(B, n, Vw, Ve, dw, de, z, min(budget_s, 12.0))
    if mt.get('value'):
        out = dict(mt)
        out['single_thread_numpy'] = port
    else:
        out = dict(port)
        out['multithreaded_error'] = mt.get('error')
    out['host_cores'] = host
    return out


def query_kernel_trace(Q, V, d, k, timeout=240):
    """Per-kernel average durations of the C5 scoring call: `rocprofv3 --kernel-trace` over this script's
    --profile-query-inner loop (the scorer runs its chunks on two streams of its own, so in-process HIP events
    would have to serialise them; the trace sees every launch as it really ran).  {kernel: {calls, avg_us}} or
    (None, reason)."""
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    import re
    import sqlite3
    tmp = tempfile.mkdtemp(prefix='sert_qtrace_')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '-d', tmp, '-o', 'q', '--', sys.executable, os.path.abspath(__file__),
               '--profile-query-inner', '--query-shape', '%d,%d,%d,%d' % (Q, V, d, k)]
        env = dict(os.environ, TMPDIR='/tmp')
        for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(key, None)
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
        found = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.endswith('.db')]
        if r.returncode != 0 or not found:
            return None, 'rocprofv3 --kernel-trace failed (rc %d): %s' % (r.returncode, r.stderr.decode()[-300:])
        db = sqlite3.connect(found[0])
        out = {}
        spans = {}
        for name, st, en in db.execute('select name, start, end from kernels order by start'):
            name = re.sub(r'\[clone .*\]', '', name).replace('sert::', '').replace('void ', '').split('(')[0].strip()
            spans.setdefault(name, []).append((st, en))
        for name, iv in spans.items():
            # the scorer runs its query chunks on two streams: launches of one kernel overlap.  avg_us = per launch as
            # traced (stretched by the neighbour); busy_us = the UNION of the launches' intervals, i.e. the time
            # during which at least one launch of this kernel was running
            busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
            for st, en in iv[1:]:
                if st > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = st, en
                else:
                    cur_e = max(cur_e, en)
            busy += cur_e - cur_s
            out[name] = {'calls': len(iv), 'avg_us': sum(e - s_ for s_, e in iv) / len(iv) / 1e3, 'busy_us': busy / 1e3}
        return out, 'rocprofv3 --kernel-trace over %d timed scoring calls' % QUERY_INNER_CALLS
    except Exception as e:   # noqa: BLE001 -- best effort: the bench line must still appear
        return None, 'kernel trace failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


QUERY_INNER_CALLS = 4


def query_inner(_capi, Q, V, d, k):
    """The workload of query_kernel_trace: the C5 scoring call, QUERY_INNER_CALLS times after one warm-up."""
    rng = np.random.RandomState(7)
    E = rng.randn(V, d).astype(np.float32)
    P = np.tanh(rng.randn(Q, d)).astype(np.float32)
    sc = _capi.Scorer(E)
    Pq = sc.query_buffer(Q)
    np.copyto(Pq, P)
    for _ in range(1 + QUERY_INNER_CALLS):
        sc.topk(Pq, k)
    sc.close()


def query_roofline(_capi, trace, Q, V, d, k, device=0):
    """`roofline` of the query record.  The dominant kernel of the call is the bf16 filter GEMM
    (score_filter_bf16<false>: every query against every entity on v_mfma_f32_32x32x16_bf16; its flops are the
    2 Q V d' of the padded bf16 operands, d' = d rounded up to 64), priced against the bf16 DENSE MFMA peak; the
    selection kernel (topk_from_groups_rescore) fetches ~(candidates with s >= s_k - 2 delta) entity rows of
    4 d bytes per query for the exact fp32 re-scoring and is priced against the measured row-fetch ceiling of a
    table of the entity table's size."""
    kp = (d + 63) // 64 * 64
    filt = [(n, v) for n, v in trace.items() if n.startswith('score_filter_bf16<false>') or n.startswith('score_filter_bf16<0>')]
    if not filt:
        filt = [(n, v) for n, v in trace.items() if n.startswith('score_filter_bf16')]
        filt = sorted(filt, key=lambda nv: -nv[1]['avg_us'])[:1]
    if not filt:
        return None
    name, rec = filt[0]
    launches_per_call = rec['calls'] / float(1 + QUERY_INNER_CALLS)
    flops = 2.0 * Q * V * kp / launches_per_call
    # the launches of one call overlap on the scorer's two streams: the rate is taken over the time during which
    # the kernel was running at all (union of its launches), per scoring call
    busy_per_call_us = rec['busy_us'] / float(1 + QUERY_INNER_CALLS)
    ach = 2.0 * Q * V * kp / (busy_per_call_us * 1e-6) / 1e12
    out = {'kernel': 'filter GEMM', 'hip_kernel': name, 'bound': 'mfma', 'achieved': round(ach, 1), 'peak': MFMA_BF16_PEAK_TFLOPS,
           'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'avg_us': round(rec['avg_us'], 2),
           'busy_us_per_call': round(busy_per_call_us, 1),
           'launches_per_call': launches_per_call, 'algorithmic_flops_per_launch': flops, 'traffic': None,
           'peak_is': 'bf16 MFMA dense peak (the filter runs on the bf16 matrix pipe; reported scores are exact fp32)',
           'note': 'the kernel is bound by its compare / ballot / rank epilogue and by resident waves, not by the matrix '
                   'pipe (DESIGN.md section 3, knock-outs): 145 us of a 260 us chunk are loads + LDS + MFMA'}
    sel = [(n, v) for n, v in trace.items() if n.startswith('topk_from_groups_rescore')]
    if sel:
        n2, r2 = sel[0]
        ceil = None
        try:
            rb = 4 * d
            us = _capi.bench_memory(_capi.MEMBENCH_GATHER, 65536 * rb, table_bytes=V * rb, row_bytes=rb, window=10, iters=10, device=device)
            ceil = 65536 * rb * 10 / (us * 1e-6) / 1e9
        except Exception:   # noqa: BLE001
            pass
        out['rescoring'] = {'hip_kernel': n2, 'avg_us': round(r2['avg_us'], 2), 'bound': 'cache',
                            'row_fetch_ceiling_GBps': round(ceil, 1) if ceil else None,
                            'ceiling_is': 'vs_gather_mean over uniformly random %d-byte rows of a %.0f MB table (sert_bench_memory)' % (4 * d, V * 4.0 * d / 1e6),
                            'note': 'reads its sparse candidate lists (~1/3 of the 32-byte groups occupied) and re-scores '
                                    '~180 of ~600 candidates per query with exact fp32 dot products'}
    out['kernels'] = {n: {'calls_per_scoring_call': v['calls'] / float(1 + QUERY_INNER_CALLS), 'avg_us': round(v['avg_us'], 2),
                          'busy_us_per_call': round(v['busy_us'] / float(1 + QUERY_INNER_CALLS), 1)}
                      for n, v in sorted(trace.items(), key=lambda nv: -nv[1]['avg_us'] * nv[1]['calls'])}
    return out


def query_bench(_capi, Q=10000, V=100000, d=128, k=100, reps=3, cpu_budget=8.0, cpu=True, trace=True):
    """BASELINE configs[4]: Q synthetic query projections x V_e entities, batched
    cosine scoring + top-k (bin/query.py:239-370).  queries/s includes the H2D of
    the projections and the D2H of the (Q,k) results; the entity table is
    resident (VectorSpaceCallback.__init__ uploads it once)."""
    rng = np.random.RandomState(7)
    E = rng.randn(V, d).astype(np.float32)
    P = np.tanh(rng.randn(Q, d)).astype(np.float32)
    sc = _capi.Scorer(E)
    sc.topk(P[:256], k)
    # the query block lives in the scorer's page-locked buffer (as VectorSpaceCallback.process_batch
    # builds it): the timed call still uploads it and downloads the (Q, k) results
    Pq = sc.query_buffer(Q)
    np.copyto(Pq, P)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        idx, val = sc.topk(Pq, k)
        best = min(best, time.perf_counter() - t0)
    idx, val = idx.copy(), val.copy()
    t0 = time.perf_counter()
    sc.topk(P, k)
    pageable = time.perf_counter() - t0
    out = {'workload': 'C5 query path: %d queries x V_e=%d, d_e=%d, top-%d (cosine, (cos+1)/2)' % (Q, V, d, k),
           'value': Q / best, 'unit': 'queries/s', 'ms_total': 1000 * best,
           'ms_total_pageable_query_array': 1000 * pageable,
           # 2 Q V d / time: what a plain scores-GEMM would have to sustain (the filter GEMM runs on
           # the bf16 matrix pipe, the reported scores are exact fp32 -- kernels_score_bf16.h)
           'equiv_gemm_tflops': 2.0 * Q * V * d / best / 1e12,
           'equiv_gemm_note': 'end-to-end rate of the whole call (H2D + kernels + D2H) expressed as a plain scores GEMM; it '
                              'exceeds the fp32 MFMA peak because the filter runs in bf16 -- the fraction of a peak is in '
                              '`roofline` (bf16 dense MFMA peak), not here'}
    if trace:
        tr, src = query_kernel_trace(Q, V, d, k)
        out['roofline'] = query_roofline(_capi, tr, Q, V, d, k) if tr else None
        out['roofline_source'] = src
    if cpu:
        from oracle import sert_oracle as O
        t0 = time.perf_counter()
        n = 0
        agree = 0
        while time.perf_counter() - t0 < cpu_budget and n < Q:
            order, _ = O.vectorspace_rank(P[n], E, top=k)
            agree += int(np.array_equal(order[:10], idx[n][:10]))
            n += 1
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': n / dt, 'unit': 'queries/s', 'cores': 1,
                               'host_cores': len(os.sched_getaffinity(0)), 'kind': 'port',
                               'sample': '%d of the %d queries (%.1f s), numpy oracle' % (n, Q, dt),
                               'top10_identical': '%d/%d' % (agree, n)}
        try:   # the multithreaded C scorer on a sample of the queries, every core the process may use
            from oracle import cpu_baseline as CB
            qs = min(Q, 512)
            CB.score_topk(E[:1000], P[:8], 10)
            t0 = time.perf_counter()
            ci = CB.score_topk(E, P[:qs], k)
            dt = time.perf_counter() - t0
            out['cpu_baseline_multithreaded'] = {
                'value': qs / dt, 'unit': 'queries/s', 'cores': len(os.sched_getaffinity(0)), 'kind': 'port',
                'sample': '%d of the %d queries (%.1f s), oracle/sert_cpu.c (OpenMP)' % (qs, Q, dt),
                'top10_identical': '%d/%d' % (int((ci[:, :10] == idx[:qs, :10]).all(axis=1).sum()), qs)}
        except Exception as e:   # noqa: BLE001
            out['cpu_baseline_multithreaded'] = {'error': repr(e)}
    sc.close()
    return out


def whole_step_record(work, kernels, seconds_per_step, extra=None):
    """Whole-step rates.  `algorithmic_GBps` = sum of the memory-side groups' SURVEY 8(d) bytes / step time --
    a rate of REQUESTED bytes, most of them served by L2 and the Infinity Cache, so it may exceed the HBM
    peak and is not reported as a fraction of it.  `counted_hbm_GBps` / `frac_of_hbm_peak` = what the PMC
    passes counted crossing the fabric for the same groups / step time (null without counters)."""
    alg = _step_bytes(work, kernels)
    counted = [v.get('hbm_bytes_pmc') for k, v in kernels.items()]
    have = [c for c in counted if c is not None]
    rec = {'algorithmic_bytes': alg, 'algorithmic_GBps': round(alg / seconds_per_step / 1e9, 1),
           'counted_hbm_bytes': (sum(have) if have else None),
           'counted_hbm_GBps': (round(sum(have) / seconds_per_step / 1e9, 1) if have else None),
           'frac_of_hbm_peak': (round(sum(have) / seconds_per_step / 1e9 / HBM_PEAK_GBS, 4) if have else None),
           'groups_with_counters': '%d of %d' % (len(have), len(counted))}
    if extra:
        rec.update(extra)
    return rec


def _step_bytes(work, present):
    """Sum of the algorithmic bytes of the memory-side groups that ran."""
    return sum(w.get('alg', 0.0) for k, w in work.items() if w['kind'] in ('stream', 'rows', 'latency') and k in present)


def measure_record(kind, models, _capi, dist, dims, steps, warmup, seed, live_pmc, num_batches=2, data=None, label=''):
    """One secondary record (loglinear / additive full softmax / C4): K untimed steps for the rate, the same
    steps with HIP events for the per-kernel table, the memory ceilings of its table shapes, and --
    live_pmc -- the two counter passes over the same workload for `traffic`."""
    B, n, Vw, Ve, d, de, z = dims['B'], dims['n'], dims['Vw'], dims['Ve'], dims['d'], dims['de'], dims['z']
    if data is None:
        rng = np.random.RandomState(seed)
        data = synth_data(rng, num_batches * B, n, Vw, Ve)
    X, y, w = data
    m = build_model(kind, models, B, n, Vw, Ve, d, de, z, X, y, w, seed=seed)
    U = None
    if kind == 'loglinear':
        U = float(np.mean([len(np.unique(X[j * B:(j + 1) * B])) for j in range(num_batches)]))
    work = group_work(kind, B, n, X.dtype.itemsize, d, de, Ve, Vw, z, U, lazy=lazy_fractions(X, B, num_batches, Vw, d))
    # (same order as the headline: ceilings, per-kernel pass, then the number -- see main)
    ceil = ceilings_for(_capi, work, device=m._engine.cfg.device)
    _, tm, _ = timed_steps(m, dist, num_batches, steps, 1, timing=True)
    dt, _, loss = timed_steps(m, dist, num_batches, steps, warmup, timing=False)
    us_in, launches_in = instep_pass(m, num_batches, max(steps, 40))
    del m
    per_kernel, source, tbg = None, None, {}
    if live_pmc:
        inner = ['--model', kind, '--batch', B, '--vocab', Vw, '--entities', Ve, '--dim', d, '--entity-dim', de,
                 '--window', n, '--negatives', z, '--num-batches', num_batches, '--seed', seed]
        per_kernel, source = pmc_traffic_live(inner, steps=5 if Vw * d > 5e7 else 13)   # (+ 3 warm-up steps: whole cycles of the lazy update's four launches)
        if per_kernel:
            tbg = traffic_by_group(per_kernel, tm, kind)
    kernels = kernel_table(tm, work, tbg)
    roof = roofline_of(kernels, tbg, source, kind=kind, instep=us_in)
    if not tbg:
        roof['traffic_note'] = source or 'counter passes not requested (--no-live-pmc)'
    rec = {
        'workload': label, 'value': steps * B / dt, 'unit': 'pairs/s', 'ms_per_step': 1000 * dt / steps, 'last_loss': loss,
        'kernels': kernels, 'roofline': roof, 'kernel_us_instep': {k: round(v, 2) for k, v in us_in.items()},
        'kernel_launches_instep': {k: round(v, 2) for k, v in launches_in.items()},
        'memory_ceilings': ceil,
        'whole_step': whole_step_record(work, kernels, dt / steps),
    }
    if U is not None:
        rec['distinct_words_per_batch'] = U
    return rec, work, tm, dt


def c4_record(models, _capi, dist, steps, live_pmc):
    """BASELINE configs[3]: V_w=500k, V_e=100k, d=300 with the reference's LSE model at the full
    batch.  The dense L2 + dense Adam of sert/models.py:764-795, :548-549 make every row of both
    tables live every step: 32 B x 180 M parameters = 5.8 GB per step is the HBM wall."""
    dims = dict(B=65536, n=10, Vw=500000, Ve=100000, d=300, de=300, z=10)
    rec, _, _, _ = measure_record(
        'vectorspace', models, _capi, dist, dims, steps, 2, 4, live_pmc,
        label='C4 LSE: VectorSpaceLanguageModel (NCE z=10, Adam) V_w=500000 V_e=100000 d=300 window=10 batch=65536')
    return rec


# ---- the ONE stdout line ------------------------------------------------------------------
LINE_LIMIT = 8000     # bytes: the driver keeps an 8 kB tail of stdout and parses the line out of it


def _r(x, nd=4):
    """Round floats for the line (6 significant digits at most carry information here)."""
    if isinstance(x, float):
        return float('%.*g' % (max(nd, 6), x))
    return x


def _short_roofline(r):
    """The dominant kernel only: the contract's keys + the counter-priced fraction."""
    if not r:
        return None
    keep = ('kernel', 'hip_kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_counter', 'frac_algorithmic',
            'achieved_algorithmic', 'traffic', 'avg_us', 'avg_us_profiled', 'traffic_over_algorithmic', 'frac_of_achievable',
            'achievable_peak', 'frac_is', 'avg_us_alone', 'frac_alone', 'frac_counter_alone')
    return {k: _r(r[k]) for k in keep if r.get(k) is not None or k == 'traffic'}


def _short_sub(rec, extra=()):
    """A sub-record in the line: value / ms_per_step / the dominant kernel's name and fractions."""
    if not rec:
        return None
    out = {k: _r(rec[k]) for k in ('value', 'unit', 'ms_per_step', 'ms_total') + tuple(extra) if rec.get(k) is not None}
    roof = rec.get('roofline')
    if roof:
        out['roofline'] = {k: _r(roof[k]) for k in ('hip_kernel', 'bound', 'frac', 'frac_counter', 'avg_us', 'avg_us_alone', 'frac_alone',
                                                    'traffic_over_algorithmic')
                           if roof.get(k) is not None}
    return out


def compact_record(full, sidecar=None):
    """The record bench.py prints: headline + `roofline` (dominant kernel) + `cpu_baseline` + one short entry per
    sub-record.  Per-kernel tables, ceilings and notes stay in the full record (`sidecar` names the file it went to).
    Always < LINE_LIMIT bytes as JSON: the optional entries are dropped from the back until it is."""
    head = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data')
    out = {k: _r(full.get(k)) for k in head}
    cfg = full.get('config') or {}
    out['config'] = {k: cfg[k] for k in ('workload', 'global_batch', 'per_gpu_batch', 'parallelism', 'id_dtype', 'lambda', 'seed',
                                          'instance_weights', 'dataset_instances', 'dataset_batches', 'upload_index_s', 'gemm')
                     if k in cfg}
    out['roofline'] = _short_roofline(full.get('roofline'))
    cb = full.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {k: _r(cb[k]) for k in ('value', 'unit', 'cores', 'kind', 'ms_per_step', 'host_cores', 'cgroup_cpu_quota')
                               if cb.get(k) is not None}
        out['cpu_baseline']['sample'] = (cb.get('sample_short') or cb.get('sample') or '')[:400]
        st = cb.get('single_thread_numpy')
        if st:
            out['cpu_baseline']['single_thread_numpy_value'] = _r(st.get('value'))
    else:
        out['cpu_baseline'] = None
    ws = full.get('whole_step') or {}
    out['whole_step'] = {k: _r(ws[k]) for k in ('counted_hbm_bytes', 'counted_hbm_GBps', 'frac_of_hbm_peak', 'kernel_us_sum_serial')
                         if ws.get(k) is not None}
    for k in ('last_loss', 'ms_per_step_instrumented', 'rccl_ranks', 'comm_bytes_per_step'):
        if full.get(k) is not None:
            out[k] = _r(full[k])
    # both readings of "scaling" at N > 1: the fixed global batch (headline) and the fixed per-GPU batch
    if full.get('weak_scaling'):
        out['strong'] = {'value': out['value'], 'ms_per_step': out['ms_per_step'], 'global_batch': cfg.get('global_batch')}
        out['weak'] = {k: _r(full['weak_scaling'].get(k)) for k in ('value', 'ms_per_step', 'global_batch', 'per_gpu_batch', 'comm_bytes_per_step')}
    optional = []
    kern = full.get('kernels') or {}
    if kern:
        optional.append(('kernel_us', {k: v['us'] for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['us'])}))
    if full.get('kernel_us_instep'):
        optional.append(('kernel_us_instep', full['kernel_us_instep']))
    if full.get('deferred_loss_readback'):
        optional.append(('deferred_loss_readback', _short_sub(full['deferred_loss_readback'])))
    if full.get('gemm_fp32_mfma_path'):
        optional.append(('gemm_fp32_mfma_path', _short_sub(full['gemm_fp32_mfma_path'])))
    if full.get('seeds'):
        sd = full['seeds']
        optional.append(('seeds', {k: (_r(v['value']) if isinstance(v, dict) else _r(v)) for k, v in sd.items()}))
    for key in ('small_batch', 'timed_mode_parity'):
        if full.get(key):
            optional.append((key, full[key]))
    for key, extra in (('loglinear', ('distinct_words_per_batch',)), ('lse_full_softmax', ()), ('c4', ())):
        if full.get(key):
            optional.append((key, _short_sub(full[key], extra)))
    if full.get('query'):
        q = _short_sub(full['query'], ('equiv_gemm_tflops',))
        for k in ('cpu_baseline', 'cpu_baseline_multithreaded'):
            c = full['query'].get(k)
            if c and c.get('value'):
                q[k] = {'value': _r(c['value']), 'cores': c.get('cores'), 'top10_identical': c.get('top10_identical')}
        optional.append(('query', q))
    dev = full.get('device')
    if dev:
        optional.append(('device', dev if len(json.dumps(dev)) < 300 else None))
    if sidecar:
        out['full_record'] = sidecar
    for k, v in optional:
        if v is not None:
            out[k] = v
    # never over the limit: drop the optional entries from the back
    names = [k for k, _ in optional]
    while len(json.dumps(out)) >= LINE_LIMIT and names:
        out.pop(names.pop(), None)
    if len(json.dumps(out)) >= LINE_LIMIT:          # (cannot happen with the keys above; keep the headline whatever it takes)
        out['cpu_baseline'] = {k: v for k, v in (out.get('cpu_baseline') or {}).items() if k != 'sample'}
        out['config'] = {'workload': str((out.get('config') or {}).get('workload'))[:300]}
    return out


def write_full_record(full):
    """The complete record (per-kernel tables, ceilings, notes): a sidecar file + stderr; returns the file's path relative to
    the repository (None where nothing is writable)."""
    text = json.dumps(full, indent=1)
    sys.stderr.write('FULL_RECORD ' + json.dumps(full) + '\n')
    for rel in (os.path.join('gpurun_out', 'bench_full.json'), os.path.join('profiles', 'bench_full_latest.json')):
        try:
            os.makedirs(os.path.join(ROOT, os.path.dirname(rel)), exist_ok=True)
            with open(os.path.join(ROOT, rel), 'w') as f:
                f.write(text)
            return rel
        except OSError:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', choices=['vectorspace', 'loglinear', 'vectorspace_softmax'], default='vectorspace')
    ap.add_argument('--batch', type=int, default=None, help='GLOBAL batch (default 65536; at --gpus N every rank takes 1/N of its rows)')
    ap.add_argument('--vocab', type=int, default=100000)
    ap.add_argument('--entities', type=int, default=1000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--entity-dim', type=int, default=None, help='d_e (default: --dim)')
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--negatives', type=int, default=10)
    ap.add_argument('--num-batches', type=int, default=64, help='data set = this many batches (SURVEY 8-d: N = 64 B)')
    ap.add_argument('--seed', type=int, default=0, help='seed of the synthetic data set (SURVEY 8-d: seeds 0..2)')
    ap.add_argument('--weights', choices=['ones', 'uniform'], default='ones',
                    help='instance weights: 1 (LSE, --no_instance_weights) or U[0.5, 2] (SURVEY 8-d second run)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=12.0)
    ap.add_argument('--no-loglinear-extra', action='store_true')
    ap.add_argument('--no-query-extra', action='store_true')
    ap.add_argument('--no-c4-extra', action='store_true')
    ap.add_argument('--no-small-extra', action='store_true', help='skip the small-batch sub-record')
    ap.add_argument('--no-seed-extra', action='store_true', help='skip the seeds 1, 2 and U[0.5, 2]-weights runs')
    ap.add_argument('--no-live-pmc', action='store_true')
    ap.add_argument('--no-weak-extra', action='store_true', help='N > 1: skip the weak-scaling sub-record')
    ap.add_argument('--profile-inner', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--profile-query-inner', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--query-shape', default='10000,100000,128,100', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # one process per GPU, started here; this process only waits for them
        from sert_amd import distributed as launcher
        sys.exit(launcher.launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    # stdout carries exactly ONE line, the JSON record: everything else that writes to
    # fd 1 (RCCL prints a version banner there at communicator creation) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from sert_amd import _build
    _build.build()
    from sert_amd import distributed as dist
    from sert_amd import models, _capi

    if args.profile_query_inner:     # the workload of query_kernel_trace (rocprofv3 --kernel-trace), nothing else
        query_inner(_capi, *[int(x) for x in args.query_shape.split(',')])
        return

    ctx = dist.init_from_env()
    if ctx.world_size != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, ctx.world_size))
    N = ctx.world_size
    kind = args.model
    # "at 1/2/4/8 GPUs with GLOBAL B fixed" -- the headline at N > 1 is the same global batch of
    # 65536 pairs split by rows over the ranks (strong scaling); the weak case is a sub-record further down
    Bg = args.batch or 65536
    if Bg % N:
        raise SystemExit('--batch %d is not divisible by --gpus %d' % (Bg, N))
    Bl = Bg // N
    n, Vw, Ve, d, z = args.window, args.vocab, args.entities, args.dim, args.negatives
    de = args.entity_dim or d

    def dataset(seed, weights, batch=None, nb=None):
        rng = np.random.RandomState(seed)
        X_, y_, w_ = synth_data(rng, (nb or args.num_batches) * (batch or Bg), n, Vw, Ve)
        if weights == 'uniform':
            w_ = rng.uniform(0.5, 2.0, len(w_)).astype(np.float32)
        return X_, y_, w_

    X, y, w = dataset(args.seed, args.weights)
    t_up = time.perf_counter()
    model = build_model(kind, models, Bg, n, Vw, Ve, d, de, z, X, y, w, seed=args.seed)
    upload_s = time.perf_counter() - t_up      # one-time, outside the timed region: id checks + per-batch inverted index (host) + H2D

    if args.profile_inner:     # the workload of the rocprofv3 --pmc passes: the steps and nothing else
        timed_steps(model, dist, args.num_batches, args.steps, args.warmup, timing=False)
        return

    # Order of the measurements.  A GPU that sat idle while the host generated the data set and built the
    # inverted index needs some tens of milliseconds of load to reach its operating clocks: the same 20 steps
    # read 319 us/step straight after the build, 304 in a second region and 295 from the third on
    # (tools/experiments/r03_step_times.py).  The bench therefore does its other GPU
    # work FIRST -- the memory ceilings of this box, then the K steps with HIP events around every
    # kernel (serialised, slower: the per-kernel table) -- and takes the number last: W untimed warm-up
    # steps, barrier + synchronise, EXACTLY K steps, synchronise + barrier.
    distinct = None
    if kind == 'loglinear':
        distinct = float(np.mean([len(np.unique(X[j * Bg:j * Bg + Bl])) for j in range(args.num_batches)]))
    work = group_work(kind, Bl, n, X.dtype.itemsize, d, de, Ve, Vw, z, distinct, shards=N,
                      lazy=lazy_fractions(X, Bg, args.num_batches, Vw, d) if N == 1 else None)
    # Memory ceilings: every rank measures them on ITS GPU (each has its own clocks to bring up) -- unless
    # several ranks share one device (the one-GPU test transport, SERT_COMM=host): stream and row-fetch rates
    # taken while N processes compete for one memory system are no ceiling of anything, and fractions against
    # them exceed 1.  Then there are none, and every fraction that needs one is null.
    my_device = model._engine.cfg.device
    devices = dist.all_gather_object((os.uname().nodename, my_device)) if N > 1 else [(os.uname().nodename, my_device)]
    shared_device = len(set(devices)) < len(devices)
    if shared_device:
        ceilings = None
    else:
        ceilings = ceilings_for(_capi, work, device=my_device)
    dt_instr, timings, _ = timed_steps(model, dist, args.num_batches, args.steps, 2, timing=True)
    dt, _, last_loss = timed_steps(model, dist, args.num_batches, args.steps, args.warmup, timing=False)
    value = args.steps * Bg / dt
    # (behind the number: the same K steps once more with every kernel's own dispatch timed IN the step)
    # (its own step count: a 20-step run holds five or six passes of the lazy update that read everything among its launches, a 100-step one
    #  a quarter -- the in-step average is taken over at least 100 steps whatever K is)
    us_instep, launches_instep = instep_pass(model, args.num_batches, max(args.steps, 100)) if N == 1 else ({}, {})

    # pass 3 (extra, not the headline): the same steps with the loss read-back deferred
    # (sert_train_batches: 25 batches per host synchronisation) -- what the per-step
    # synchronisation of the reference's epoch loop costs
    eng = model._engine
    eng.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        k = min(25, args.steps - done)
        eng.train_batches([(done + i) % args.num_batches for i in range(k)])
        done += k
    eng.synchronize()
    dist.barrier()
    dt_async = dist.all_reduce_max(time.perf_counter() - t0)
    # pass 4 (extra): the same W + K steps with every contraction on the fp32 MFMA kernels (SERT_GEMM_FP32=1 is read at
    # every launch) -- the number of the path whose products are single fp32 MFMA instructions, beside the headline
    gemm_fp32 = None
    if kind == 'vectorspace' and os.environ.get('SERT_GEMM_FP32', '0') in ('', '0'):
        os.environ['SERT_GEMM_FP32'] = '1'
        try:
            dt32, _, loss32 = timed_steps(model, dist, args.num_batches, args.steps, args.warmup, timing=False)
        finally:
            os.environ['SERT_GEMM_FP32'] = '0'
        gemm_fp32 = {'value': args.steps * Bg / dt32, 'unit': 'pairs/s', 'ms_per_step': 1000.0 * dt32 / args.steps,
                     'last_loss': loss32,
                     'note': 'SERT_GEMM_FP32=1: the three GEMMs of the step as v_mfma_f32_32x32x2_f32 (gemm.h) instead of six '
                             'bf16 MFMA products of exactly split operands (gemm_x3.h); same W warm-up + K timed steps'}
    device_info = _capi.device_info(model._engine.cfg.device)
    comm = getattr(model, 'comm_info', lambda: None)()
    del model, eng

    # seeds 0..2 and a second run with instance weights w ~ U[0.5, 2] (same K steps each)
    seed_runs = None
    if N == 1 and not args.no_seed_extra:
        seed_runs = {'seed_%d' % args.seed: {'value': value, 'ms_per_step': 1000.0 * dt / args.steps}}
        nbs = min(args.num_batches, 8)     # (8 batches each: bounds the bench's wall time; the headline runs the full data set)
        for sd, wt in ((1, 'ones'), (2, 'ones'), (args.seed, 'uniform')):
            Xs, ys, ws = dataset(sd, wt, nb=nbs)
            ms_ = build_model(kind, models, Bg, n, Vw, Ve, d, de, z, Xs, ys, ws, seed=sd)
            timed_steps(ms_, dist, nbs, args.steps, 2, timing=True)     # (GPU idle during the build: as above)
            dts, _, _ = timed_steps(ms_, dist, nbs, args.steps, args.warmup, timing=False)
            seed_runs['seed_%d%s' % (sd, '_w_uniform_0.5_2' if wt == 'uniform' else '')] = {
                'value': args.steps * Bg / dts, 'ms_per_step': 1000.0 * dts / args.steps}
            del ms_

    # N > 1, sub-record: weak scaling -- 65536 rows PER GPU, global batch N x 65536 (the exchange grows with the
    # rows a batch touches, sub-linearly, so this is the friendlier case; it is NOT BASELINE configs[2])
    weak = None
    if N > 1 and not args.no_weak_extra:
        nbw = min(args.num_batches, 8)
        Xw, yw, ww = dataset(args.seed, args.weights, Bg * N, nb=nbw)
        ms = build_model(kind, models, Bg * N, n, Vw, Ve, d, de, z, Xw, yw, ww, seed=args.seed)
        timed_steps(ms, dist, nbw, args.steps, 2, timing=True)
        dts, _, _ = timed_steps(ms, dist, nbw, args.steps, args.warmup, timing=False)
        cw = getattr(ms, 'comm_info', lambda: None)() or {}
        weak = {'value': args.steps * Bg * N / dts, 'unit': 'pairs/s', 'ms_per_step': 1000.0 * dts / args.steps,
                'scaling': 'weak', 'global_batch': Bg * N, 'per_gpu_batch': Bg,
                'comm_bytes_per_step': cw.get('comm_bytes_per_step'),
                'note': 'weak scaling: every GPU keeps the single-GPU batch of %d rows; not the configuration '
                        'BASELINE.json configs[2] names (same LSE config = same global batch)' % Bg}
        del ms, Xw, yw, ww

    out = None
    s = X.dtype.itemsize
    live = not args.no_live_pmc
    if ctx.rank == 0:
        per_kernel, traffic_source = (None, None)
        if N == 1 and live:
            inner = ['--model', kind, '--batch', Bl, '--vocab', Vw, '--entities', Ve, '--dim', d, '--entity-dim', de,
                     '--window', n, '--negatives', z, '--num-batches', min(args.num_batches, 16), '--seed', args.seed,
                     '--weights', args.weights]
            per_kernel, traffic_source = pmc_traffic_live(inner)
        if per_kernel is None and kind == 'vectorspace' and Bl == 65536 and os.path.exists(os.path.join(ROOT, COMMITTED_PMC)):
            why = traffic_source
            with open(os.path.join(ROOT, COMMITTED_PMC)) as f:
                per_kernel = json.load(f)
            traffic_source = COMMITTED_PMC + ' (committed earlier; live passes unavailable: %s)' % why
        tbg = traffic_by_group(per_kernel, timings, kind) if per_kernel else {}
        kernels = kernel_table(timings, work, tbg)
        roofline = roofline_of(kernels, tbg, traffic_source, kind=kind, instep=us_instep if N == 1 else None)
        if per_kernel is None or not tbg:
            roofline['traffic_note'] = traffic_source
        step_flops = sum(w_['flops'] for k, w_ in work.items() if w_['kind'] == 'mfma' and k in kernels)
        kernel_sum_us = sum(v['us'] for v in kernels.values())
        out = {
            'metric': 'training_pairs_per_sec', 'value': value, 'unit': 'pairs/s',
            'n_gpus': N, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000.0 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': ('C2 LSE: %s V_w=%d V_e=%d d=%d window=%d batch=%d%s' % (
                    {'vectorspace': 'VectorSpaceLanguageModel (NCE z=%d, Adam)' % z,
                     'loglinear': 'LanguageModel (full softmax, Adadelta)',
                     'vectorspace_softmax': 'VectorSpaceSoftmaxLanguageModel (additive full softmax, Adam)'}[kind],
                    Vw, Ve, d, n, Bg, ('' if de == d else ' d_e=%d' % de) + ('' if N == 1 else ' (global; %d rows per GPU)' % Bl))),
                'global_batch': Bg, 'per_gpu_batch': Bl,
                'parallelism': 'dp%d' % N if N == 1 else 'dp%d, %s' % (N, (comm or {}).get('exchange', 'data parallel')),
                'id_dtype': str(X.dtype), 'lambda': 0.01, 'seed': args.seed,
                'dataset_instances': int(len(X)), 'dataset_batches': int(len(X) // Bg), 'upload_index_s': round(upload_s, 3),
                'instance_weights': '1' if args.weights == 'ones' else 'U[0.5, 2]',
                'gemm_arithmetic': ('fp32 MFMA (SERT_GEMM_FP32=1)' if os.environ.get('SERT_GEMM_FP32', '0') not in ('', '0') else
                                    'fp32 operands, fp32 accumulators, fp32 results; each operand is split exactly into three bf16 '
                                    'pieces and a product runs as six bf16 MFMA products (gemm_x3.h): against float64 the error is '
                                    'that of the fp32 accumulation, as on the fp32 MFMA kernels (tests/test_gpu_gemm.py, same bounds)'),
            },
            'roofline': roofline,
            'memory_ceilings': (dict(ceilings, note='measured in this process on this box (sert_bench_memory): float4 stream copy / '
                                     'read over 1.2 GB, vs_gather_mean over uniformly random rows (window 10), adam_l2 over '
                                     'four separately allocated arrays; spec HBM peak %.0f GB/s' % HBM_PEAK_GBS)
                                if ceilings is not None else None),
            'whole_step': whole_step_record(work, kernels, dt / args.steps, {
                'algorithmic_flops': step_flops, 'kernel_us_sum_serial': round(kernel_sum_us, 1)}),
            'ms_per_step_instrumented': 1000.0 * dt_instr / args.steps,
            'measurement_order': 'memory ceilings, then the K steps with per-kernel HIP events, then the number '
                                 '(%d untimed steps that bring the clocks up, W untimed warm-up steps, exactly K timed steps '
                                 'between barrier + synchronise): the GPU is at operating clocks when the timed region starts' % CLOCK_STEPS,
            'deferred_loss_readback': {'value': args.steps * Bg / dt_async, 'unit': 'pairs/s',
                                       'ms_per_step': 1000.0 * dt_async / args.steps,
                                       'note': '25 batches per host synchronisation (additive mode; the headline '
                                               'value keeps the per-step read-back of the reference loop)'},
            'kernels': kernels,
            'kernel_us_instep': {k: round(v, 2) for k, v in us_instep.items()},
            'kernel_launches_instep': {k: round(v, 2) for k, v in launches_instep.items()},
            'last_loss': last_loss,
            'device': device_info,
        }
        if gemm_fp32 is not None:
            out['gemm_fp32_mfma_path'] = gemm_fp32
        if seed_runs is not None:
            vals = [v['value'] for k, v in seed_runs.items() if 'uniform' not in k]
            out['seeds'] = dict(seed_runs, mean_value_seeds_0_1_2=float(np.mean(vals)),
                                spread_rel=float((max(vals) - min(vals)) / np.mean(vals)))
        if shared_device:
            out['memory_ceilings_note'] = ('%d ranks share one device (test transport): no ceilings were measured; per-kernel '
                                           'fractions that need one are null, times are those of kernels competing for one GPU' % N)
        if comm:
            if comm.get('transport') == 'rccl' and comm.get('rccl_ranks') != N:
                raise SystemExit('RCCL communicator spans %r ranks, --gpus %d' % (comm.get('rccl_ranks'), N))
            out['rccl_ranks'] = comm.get('rccl_ranks')
            out['comm_bytes_per_step'] = comm.get('comm_bytes_per_step')
            out['comm'] = comm
        if weak is not None:
            out['weak_scaling'] = weak

    # extra: the reference's full-softmax model (loglinear) at the C2 dims AND the C2 batch
    if ctx.rank == 0 and N == 1 and kind == 'vectorspace' and not args.no_loglinear_extra:
        st = max(5, min(20, args.steps))
        dims = dict(B=65536, n=n, Vw=Vw, Ve=Ve, d=d, de=d, z=z)
        rec, work2, _, dt2 = measure_record(
            'loglinear', models, _capi, dist, dims, st, 3, 1, live,
            label='LanguageModel (full softmax over V_e, Adadelta) V_w=%d V_e=%d d=%d window=%d batch=65536' % (Vw, Ve, d, n))
        fl_exec = sum(w_['flops'] for w_ in work2.values() if w_['kind'] == 'mfma')
        rec['mfma_tflops_executed'] = fl_exec / (dt2 / st) / 1e12
        rec['note'] = ('the three GEMMs run on the %.0f distinct words of a batch of %d tokens (duplicate tokens share '
                       'their logit row); mfma_tflops_executed counts those flops only' % (rec['distinct_words_per_batch'], 65536 * n))
        out['loglinear'] = rec
        # extra: the ADDITIVE full-softmax LSE variant (BASELINE.json configs[1] wording:
        # "embed gather + MFMA projection + full softmax"), same dims and batch as the headline
        dims = dict(B=Bl, n=n, Vw=Vw, Ve=Ve, d=d, de=d, z=z)
        rec, _, _, dt3 = measure_record(
            'vectorspace_softmax', models, _capi, dist, dims, st, 3, 2, live, num_batches=4,
            label='VectorSpaceSoftmaxLanguageModel (additive, not in the reference): gather + mean-pool + tanh projection + '
                  'full softmax over V_e=%d, Adam; V_w=%d d=%d window=%d batch=%d' % (Ve, Vw, d, n, Bl))
        rec['mfma_tflops_whole_step'] = (6.0 * Bl * d * d + 6.0 * Bl * d * Ve) / (dt3 / st) / 1e12
        rec['parity'] = ('self-checked: this model has no counterpart in the reference (SURVEY 8-a12); its only oracle is the '
                         "builder's own restatement + finite differences (tests/test_gpu_fullbatch.py at this batch)")
        out['lse_full_softmax'] = rec

    if ctx.rank == 0 and N == 1 and kind == 'vectorspace' and not args.no_c4_extra:
        out['c4'] = c4_record(models, _capi, dist, max(5, min(10, args.steps)), live)

    # extra: the step at the reference's OWN batch sizes (its canonical hyper-parameters) and at the per-GPU batch of the
    # 8-GPU strong-scaling case -- latency-bound steps; ms per step only (same loop: hints, per-step loss read-back)
    if ctx.rank == 0 and N == 1 and kind == 'vectorspace' and not args.no_small_extra and not args.no_c4_extra:   # (quick runs skip both)
        small = {}
        for name, kd, c in (
                ('c2_dims_batch_8192', 'vectorspace', dict(B=8192, n=n, Vw=Vw, Ve=Ve, d=d, de=de)),
                ('product_search_settings_batch_4096_dw300_de128_Ve32768', 'vectorspace', dict(B=4096, n=10, Vw=100000, Ve=32768, d=300, de=128)),
                ('w3c_loglinear_settings_batch_1024_d300_Ve715', 'loglinear', dict(B=1024, n=8, Vw=100000, Ve=715, d=300, de=300))):
            rng_ = np.random.RandomState(5)
            Xs, ys, ws = synth_data(rng_, 8 * c['B'], c['n'], c['Vw'], c['Ve'])
            ms_ = build_model(kd, models, c['B'], c['n'], c['Vw'], c['Ve'], c['d'], c['de'], z, Xs, ys, ws, seed=5)
            st_ = max(50, min(200, 10 * args.steps))
            dts, _, _ = timed_steps(ms_, dist, 8, st_, 10, timing=False)
            small[name] = {'ms_per_step': round(1000.0 * dts / st_, 4), 'value': round(st_ * c['B'] / dts, 0)}
            del ms_
        out['small_batch'] = small

    if ctx.rank == 0 and N == 1 and not args.no_query_extra:
        out['query'] = query_bench(_capi, cpu=not args.no_cpu_baseline, trace=live)

    if ctx.rank == 0 and N == 1 and not args.no_cpu_baseline and kind == 'vectorspace':
        out['cpu_baseline'] = cpu_baseline(Bl, n, Vw, Ve, d, de, z, args.cpu_budget)
    elif ctx.rank == 0:
        out['cpu_baseline'] = None

    if ctx.rank == 0:
        sys.stdout.flush()
        line = json.dumps(compact_record(out, sidecar=write_full_record(out)))
        assert len(line) < LINE_LIMIT, len(line)
        os.write(json_fd, (line + '\n').encode())
    dist.barrier()
    dist.shutdown()


if __name__ == '__main__':
    main()
