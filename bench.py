#!/usr/bin/env python
"""bench.py -- training (word-window, entity) pairs/s of the SERT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one call of train_fn(batch_index) (sert/models.py:581-588): forward,
backward, L2, dense optimiser update on one batch of synthetic (window, entity)
pairs, INCLUDING the per-step loss read-back the reference's epoch loop performs
(sert/models.py:369-379).  Workload at N=1: BASELINE.json configs[1] -- the LSE
config |V_w|=100k, |V_e|=1k, d=128, window=10, batch=65536 -- run with the
reference's LSE model (VectorSpaceLanguageModel: NCE, z=10, Adam).  N>1: weak
scaling, one process per GPU (launched by torch.distributed.run), per-GPU batch
fixed at 65536, gradients summed by one RCCL all-reduce per step.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel group
(HIP events on the model's stream, measured live in the timed region),
`cpu_baseline` the oracle (numpy restatement of the reference graph) timed on
this node's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA dense peak


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


def group_work(kind, B, n, s, dw, de, Ve, Vw, z):
    """Algorithmic bytes / flops per timed kernel (SURVEY 8(d) per-pair figures x
    the pairs one step processes; per-step optimiser term 32 B per parameter).
    Keys = timing group names of libsert_hip.so (sert_timing_name)."""
    if kind == 'vectorspace':
        return {
            'gather':               ('hbm', B * (n * s + 4 * n * dw)),          # vs_gather_mean
            'gemm_fwd':             ('mfma', 2.0 * B * dw * de),               # gemm_f32_mfma NN+tanh
            'loss':                 ('hbm', B * (8 + 4 * (1 + z) * de)),       # vs_nce
            'entity_sort':          ('hbm', B * (1 + z) * 16.0),               # csort_* (keys+values r/w)
            'entity_grad_reduce':   ('hbm', B * (8 * (1 + z) * de)),           # egrad_chunk_reduce
            'entity_grad_fixup':    ('hbm', 8.0 * Ve * de),                    # egrad_fixup
            'gemm_dW':              ('mfma', 2.0 * B * dw * de),               # gemm_f32_mfma TN split-K
            'splitk_combine':       ('hbm', 4.0 * 1024 * (dw * de + de)),      # reduce_partials
            'gemm_dX':              ('mfma', 2.0 * B * dw * de),               # gemm_f32_mfma NT
            'word_grad_segsum':     ('hbm', B * (8 * n * dw)),                 # segsum_rows
            'optimizer_word_table': ('hbm', 32.0 * Vw * dw),                   # adam_l2 (R_w)
            'optimizer_other':      ('hbm', 32.0 * (Ve * de + dw * de + de)),  # adam_l2 (R_e, W, b)
        }
    return {
        'gather':               ('hbm', B * (n * s + 4 * n * dw)),
        'gemm_fwd':             ('mfma', 2.0 * B * n * dw * Ve),
        'loss':                 ('hbm', B * (8.0 * n * Ve)),                     # Z read + dZ write
        'gemm_dW':              ('mfma', 2.0 * B * n * dw * Ve),
        'gemm_dX':              ('mfma', 2.0 * B * n * dw * Ve),
        'word_grad_segsum':     ('hbm', B * (8 * n * dw)),
        'optimizer_word_table': ('hbm', 32.0 * Vw * dw),
        'optimizer_other':      ('hbm', 32.0 * (dw * Ve + Ve)),
    }


KERNEL_OF_GROUP = {
    'gather': 'vs_gather_mean<unsigned int, 4>', 'gemm_fwd': 'gemm_f32_mfma<false, false, 2, false, true, true>',
    'loss': 'vs_nce<2, true>', 'entity_grad_reduce': 'egrad_chunk_reduce<4, 2>',
    'entity_grad_fixup': 'egrad_fixup<4>', 'gemm_dW': 'gemm_f32_mfma<true, false, 0, true, true, true>',
    'splitk_combine': 'reduce_partials', 'gemm_dX': 'gemm_f32_mfma<false, true, 0, false, true, true>',
    'word_grad_segsum': 'segsum_rows<32, false, false>', 'optimizer_word_table': 'adam_l2<false>',
    'optimizer_other': 'adam_l2<false>', 'entity_sort': 'csort_scatter',
}
PMC_FILE = 'profiles/r01_h_vs_c2_pmc.json'


def load_pmc():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2
    per the gfx950 correction + WRITE_SIZE; tools/rocpd_pmc.py)."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def pmc_traffic(pmc, hip_kernel, avg_us):
    """HBM bytes of the profiled launch of `hip_kernel` whose duration is closest to
    the measured one (the PMC file is keyed 'kernel @grid_size')."""
    best = None
    for key, rec in pmc.items():
        if hip_kernel and key.split(' @')[0].startswith(hip_kernel.split('(')[0]):
            d = abs(rec.get('avg_us_profiled', 0.0) - avg_us)
            if best is None or d < best[0]:
                best = (d, rec.get('hbm_bytes'))
    return best[1] if best else None


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


def timed_steps(model, dist, num_batches, steps, warmup, timing=True):
    eng = model._engine
    for i in range(warmup):
        model.train_fn(i % num_batches)
    eng.timing_reset()
    eng.timing_enable(timing)
    eng.synchronize()
    dist.barrier()
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
    eng.timing_enable(False)
    if not np.isfinite(last):
        raise RuntimeError('non-finite loss in the timed region')
    timings = eng.timings()
    if model._engine.cfg.kind == 0 and 'entity_grad_reduce' in timings:
        # loglinear reuses this timing slot for the per-distinct-word sums of dJ / r
        timings['per_word_dz_sums'] = timings.pop('entity_grad_reduce')
    return dist.all_reduce_max(dt), timings, float(last)


def cpu_baseline(kind, B, n, Vw, Ve, dw, de, z, budget_s=15.0):
    """The oracle (numpy restatement of the reference graph -- NOT Theano, which
    cannot run here) timed on this node's host cores."""
    from oracle import sert_oracle as O
    rng = np.random.RandomState(123)
    X, y, w = synth_data(rng, B, n, Vw, Ve)
    Rw = glorot(rng, (Vw, dw))
    if kind == 'vectorspace':
        ora = O.VectorSpaceOracle(B, n, z, Rw, glorot(rng, (Ve, de)), glorot(rng, (dw, de)),
                                  np.zeros(de, np.float32), 0.01)
        step = lambda: ora.train_step(X, y, w, rng.randint(0, Ve, size=(B, z)))
    else:
        ora = O.LogLinearOracle(B, n, Rw, glorot(rng, (dw, Ve)), np.zeros(Ve, np.float32), 0.01)
        step = lambda: ora.train_step(X, y, w)
    t0 = time.perf_counter()
    steps = 0
    while True:
        step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= 8:
            break
    # cores: what the port really uses.  Its time goes to single-threaded NumPy kernels
    # (fancy-index gather, ufunc.at scatter-add, elementwise optimiser); only the two small
    # projections call the multi-threaded BLAS (< 2 % of a step) -- like the reference's
    # Theano CPU path (C loops + BLAS).  host_cores = what the node offers.
    return dict(value=steps * B / dt, unit='pairs/s', cores=1,
                host_cores=len(os.sched_getaffinity(0)), kind='port',
                sample='%d steps of B=%d (%d pairs, %.1f s) of the same workload; NumPy restatement of '
                       'the reference graph (oracle/), not Theano; effectively single-threaded '
                       '(BLAS-threaded matmuls are < 2 %% of the step)' %
                       (steps, B, steps * B, dt))


def query_bench(_capi, Q=10000, V=100000, d=128, k=100, reps=3, cpu_budget=8.0, cpu=True):
    """BASELINE configs[4]: Q synthetic query projections x V_e entities, batched
    cosine scoring + top-k (bin/query.py:239-370).  queries/s includes the H2D of
    the projections and the D2H of the (Q,k) results; the entity table is
    resident (VectorSpaceCallback.__init__ uploads it once)."""
    rng = np.random.RandomState(7)
    E = rng.randn(V, d).astype(np.float32)
    P = np.tanh(rng.randn(Q, d)).astype(np.float32)
    sc = _capi.Scorer(E)
    sc.topk(P[:256], k)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        idx, val = sc.topk(P, k)
        best = min(best, time.perf_counter() - t0)
    out = {'workload': 'C5 query path: %d queries x V_e=%d, d_e=%d, top-%d (cosine, (cos+1)/2)' % (Q, V, d, k),
           'value': Q / best, 'unit': 'queries/s', 'ms_total': 1000 * best,
           # 2 Q V d / time: what a plain scores-GEMM would have to sustain (the filter GEMM runs on
           # the bf16 matrix pipe, the reported scores are exact fp32 -- kernels_score_bf16.h)
           'equiv_gemm_tflops': 2.0 * Q * V * d / best / 1e12}
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
                               'host_cores': len(os.sched_getaffinity(0)), 'kind': 'port', 'sample': '%d of the %d queries (%.1f s), numpy oracle' % (n, Q, dt),
                               'top10_identical': '%d/%d' % (agree, n)}
    sc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', choices=['vectorspace', 'loglinear'], default='vectorspace')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 65536; loglinear 8192)')
    ap.add_argument('--vocab', type=int, default=100000)
    ap.add_argument('--entities', type=int, default=1000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--entity-dim', type=int, default=None, help='d_e (default: --dim)')
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--negatives', type=int, default=10)
    ap.add_argument('--num-batches', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=15.0)
    ap.add_argument('--no-loglinear-extra', action='store_true')
    ap.add_argument('--no-query-extra', action='store_true')
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: everything else that writes to
    # fd 1 (RCCL prints a version banner there at communicator creation) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from sert_amd import _build
    _build.build()
    from sert_amd import distributed as dist
    from sert_amd import models, _capi

    ctx = dist.init_from_env()
    if ctx.world_size != args.gpus and not (args.gpus == 1 and ctx.world_size == 1):
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run '
                         '--nproc-per-node %d' % (args.gpus, ctx.world_size, args.gpus))
    N = ctx.world_size
    kind = args.model
    Bl = args.batch or (65536 if kind == 'vectorspace' else 8192)
    Bg = Bl * N
    n, Vw, Ve, d, z = args.window, args.vocab, args.entities, args.dim, args.negatives
    de = args.entity_dim or d
    rng = np.random.RandomState(0)
    X, y, w = synth_data(rng, args.num_batches * Bg, n, Vw, Ve)
    model = build_model(kind, models, Bg, n, Vw, Ve, d, de, z, X, y, w, seed=0)

    # pass 1 (the number): K untimed steps.  pass 2: the same K steps again with HIP
    # events around every kernel (serialised, slower) for the per-kernel table.
    dt, _, last_loss = timed_steps(model, dist, args.num_batches, args.steps, args.warmup, timing=False)
    value = args.steps * Bg / dt
    dt_instr, timings, _ = timed_steps(model, dist, args.num_batches, args.steps, 2, timing=True)

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

    out = None
    if ctx.rank == 0:
        s = X.dtype.itemsize
        work = group_work(kind, Bl, n, s, d, de, Ve, Vw, z)
        kernels = {}
        for name, us in timings.items():
            if us <= 0 or name not in work:
                continue
            bound, amount = work[name]
            if bound == 'hbm':
                ach = amount / (us * 1e-6) / 1e9
                kernels[name] = dict(us=round(us, 2), bound='hbm', achieved=round(ach, 1),
                                     unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4))
            else:
                ach = amount / (us * 1e-6) / 1e12
                kernels[name] = dict(us=round(us, 2), bound='mfma', achieved=round(ach, 2),
                                     unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4))
        for name, us in timings.items():
            if us > 0 and name not in kernels:
                kernels[name] = dict(us=round(us, 2))
        pmc = load_pmc()
        dom = max((k for k in kernels if 'bound' in kernels[k]), key=lambda k: kernels[k]['us'])
        # three launches tie at C2 (entity/word gradient reductions and the word-table
        # optimiser, 59-61 us each, order varying run to run): among kernels within 5 % of the
        # longest, report the one
        # that streams the most real HBM bytes (the PMC file) -- the reductions gather
        # cache-resident rows, their algorithmic rate is not an HBM rate
        ties = [k for k in kernels if 'bound' in kernels[k] and kernels[k]['us'] >= 0.95 * kernels[dom]['us']]
        if len(ties) > 1 and pmc:
            hbm_of = lambda k: (pmc_traffic(pmc, KERNEL_OF_GROUP.get(k), kernels[k]['us']) or 0.0)
            # prefer the kernel whose PMC bytes are closest to (not far above) its algorithmic bytes
            dom = min(ties, key=lambda k: abs(1.0 - hbm_of(k) / max(1.0, work[k][1])))
        kd = kernels[dom]
        roofline = dict(kernel=dom, hip_kernel=KERNEL_OF_GROUP.get(dom), bound=kd['bound'],
                        achieved=kd['achieved'],
                        peak=HBM_PEAK_GBS if kd['bound'] == 'hbm' else MFMA_F32_PEAK_TFLOPS,
                        unit=kd['unit'], frac=kd['frac'],
                        traffic=pmc_traffic(pmc, KERNEL_OF_GROUP.get(dom), kd['us']) if kind == 'vectorspace' else None,
                        traffic_source=PMC_FILE if pmc else None, avg_us=kd['us'],
                        note='achieved = algorithmic bytes (SURVEY 8d) / HIP-event time; tables that fit the '
                             '256 MB Infinity Cache can exceed the HBM peak')
        out = {
            'metric': 'training_pairs_per_sec', 'value': value, 'unit': 'pairs/s',
            'n_gpus': N, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000.0 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': ('C2 LSE: %s V_w=%d V_e=%d d=%d window=%d batch/GPU=%d%s' % (
                    'VectorSpaceLanguageModel (NCE z=%d, Adam)' % z if kind == 'vectorspace'
                    else 'LanguageModel (full softmax, Adadelta)',
                    Vw, Ve, d, n, Bl, ('' if de == d else ' d_e=%d' % de) + ('' if N == 1 else ' global_batch=%d' % Bg))),
                'global_batch': Bg, 'parallelism': 'dp%d' % N,
                'id_dtype': str(X.dtype), 'lambda': 0.01,
            },
            'roofline': roofline,
            'ms_per_step_instrumented': 1000.0 * dt_instr / args.steps,
            'deferred_loss_readback': {'value': args.steps * Bg / dt_async, 'unit': 'pairs/s',
                                       'ms_per_step': 1000.0 * dt_async / args.steps,
                                       'note': '25 batches per host synchronisation (additive mode; the headline '
                                               'value keeps the per-step read-back of the reference loop)'},
            'kernels': kernels,
            'last_loss': last_loss,
            'device': _capi.device_info(model._engine.cfg.device),
        }

    # extra: the reference's full-softmax model (loglinear) at the same dims
    if N == 1 and kind == 'vectorspace' and not args.no_loglinear_extra:
        del model
        Bll = 8192
        rng2 = np.random.RandomState(1)
        X2, y2, w2 = synth_data(rng2, 4 * Bll, n, Vw, Ve)
        m2 = build_model('loglinear', models, Bll, n, Vw, Ve, d, d, z, X2, y2, w2, seed=1)
        st = max(5, min(20, args.steps))
        dt2, _, _ = timed_steps(m2, dist, 4, st, 3, timing=False)
        _, tm2, _ = timed_steps(m2, dist, 4, st, 1, timing=True)
        work2 = group_work('loglinear', Bll, n, X2.dtype.itemsize, d, d, Ve, Vw, z)
        fl = sum(v for k, (b, v) in work2.items() if b == 'mfma')
        out['loglinear'] = {
            'workload': 'LanguageModel (full softmax over V_e, Adadelta) V_w=%d V_e=%d d=%d window=%d batch=%d' % (Vw, Ve, d, n, Bll),
            'value': st * Bll / dt2, 'unit': 'pairs/s', 'ms_per_step': 1000 * dt2 / st,
            'kernels_us': {k: round(v, 1) for k, v in tm2.items() if v > 0},
            'mfma_tflops_whole_step': fl / (dt2 / st) / 1e12,
        }
        del m2
        # extra: the ADDITIVE full-softmax LSE variant (BASELINE.json configs[1] wording:
        # "embed gather + MFMA projection + full softmax"), same dims and batch as the headline
        m3 = build_model('vectorspace_softmax', models, Bl, n, Vw, Ve, d, d, z, X[:4 * Bl], y[:4 * Bl],
                         w[:4 * Bl], seed=2)
        dt3, _, _ = timed_steps(m3, dist, 4, st, 3, timing=False)
        _, tm3, _ = timed_steps(m3, dist, 4, st, 1, timing=True)
        fl3 = 6.0 * Bl * d * d + 6.0 * Bl * d * Ve
        out['lse_full_softmax'] = {
            'workload': 'VectorSpaceSoftmaxLanguageModel (additive, not in the reference): gather + mean-pool + '
                        'tanh projection + full softmax over V_e=%d, Adam; V_w=%d d=%d window=%d batch=%d' % (
                            Ve, Vw, d, n, Bl),
            'value': st * Bl / dt3, 'unit': 'pairs/s', 'ms_per_step': 1000 * dt3 / st,
            'kernels_us': {k: round(v, 1) for k, v in tm3.items() if v > 0},
            'mfma_tflops_whole_step': fl3 / (dt3 / st) / 1e12,
        }
        del m3

    if ctx.rank == 0 and N == 1 and not args.no_query_extra:
        out['query'] = query_bench(_capi, cpu=not args.no_cpu_baseline)

    if ctx.rank == 0 and N == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(kind, Bl, n, Vw, Ve, d, de, z, args.cpu_budget)
    elif ctx.rank == 0:
        out['cpu_baseline'] = None

    if ctx.rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    dist.barrier()
    dist.shutdown()


if __name__ == '__main__':
    main()
