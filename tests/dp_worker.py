"""One rank of a data-parallel run on a SHARED GPU (host-mediated exchange,
SERT_COMM=host), and the same run single-process -- used by
test_gpu_models.py::test_two_ranks_on_one_gpu_match_single_process.

    python -m sert_amd.distributed 2 tests/dp_worker.py KIND OUT.npz
    (or any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE, e.g. torch.distributed.run)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_c2(kind):
    """Three steps at the headline size (C2), evaluation of one batch, the tables and the word table's
    optimiser state -- for any world size (batch_size is the GLOBAL batch)."""
    import bench
    from sert_amd import models, _capi as C
    B, n, z, Vw, Ve, d = 65536, 10, 10, 100000, 1000, 128
    rng = np.random.RandomState(0)
    X, y, w = bench.synth_data(rng, 2 * B, n, Vw, Ve)
    w = rng.uniform(0.5, 2.0, len(w)).astype(np.float32)
    models.VectorSpaceLanguageModel.sampler_seed = 1234
    m = bench.build_model('vectorspace', models, B, n, Vw, Ve, d, d, z, X, y, w, seed=0)
    out = {}
    m._engine.hint_next_batch(1)
    out['loss0'] = np.float64(m.train_fn(0))
    m._engine.hint_next_batch(0)
    out['loss1'] = np.float64(m.train_fn(1))
    out['loss2'] = np.float64(m.train_fn(0))
    out['eval0'] = np.float64(m.test_fn(0))
    for name, which in (('Rw', C.T_RW), ('Re', C.T_RE), ('W', C.T_W), ('b', C.T_B),
                        ('opt_state0_rw', C.T_STATE0_RW), ('opt_state1_rw', C.T_STATE1_RW)):
        out[name] = m._engine.get_tensor(which).copy()
    info = m.comm_info()
    out['exchange'] = np.str_(info['exchange_kind'] if info else 'none')
    out['comm_bytes_per_step'] = np.float64(info['comm_bytes_per_step'] if info else 0.0)
    out['zero1_bytes_per_step'] = np.float64(info['zero1_comm_bytes_per_step'] if info else 0.0)
    out['transport'] = np.str_(info['transport'] if info else 'none')
    out['rccl_ranks'] = np.int64(info['rccl_ranks'] if info else 0)
    out['rccl_lib'] = np.str_(loaded_rccl())
    return out


def loaded_rccl():
    """Path of the librccl mapping of this process ('' if none): lets a test prove WHICH library served it."""
    try:
        with open('/proc/self/maps') as f:
            libs = sorted(set(l.split()[-1] for l in f if 'librccl' in l))
    except OSError:
        return ''
    return ';'.join(libs)


def run_soak(steps=200):
    """`steps` hinted training steps over six batches with an evaluation pass every 50 -- the steady-state
    schedule of an epoch loop (parameters of the next batch fetched behind the owned rows' update, gradient rows
    returned beside dW, the small all-reduce, the collective all-gather in front of every evaluation), long
    enough for a lost event or a crossed collective to show as a hang or a drift."""
    from sert_amd import models, _capi as C
    from tests import util as U
    B, n, z, Vw, Ve, d = 256, 4, 5, 3000, 40, 32
    p = U.make_vs_problem(71, B * 6, n, z, Vw, Ve, d, d, zipf=True)
    np.random.seed(5)
    models.VectorSpaceLanguageModel.sampler_seed = 99
    m = models.VectorSpaceLanguageModel(
        batch_size=B, window_size=n, num_negative_samples=z, representations_init=p['Rw'],
        entity_representations_init=p['Re'], regularization_lambda=0.01,
        training_set=(p['X'], p['y'], p['w']), validation_set=(p['X'][:B * 2], p['y'][:B * 2]))
    out, losses = {}, []
    for s in range(steps):
        m._engine.hint_next_batch((s + 1) % 6 if s + 1 < steps else None)
        losses.append(m.train_fn(s % 6))
        if s % 50 == 49:
            losses.append(m.validation_error()[0])
    out['losses'] = np.array(losses, dtype=np.float64)
    for name, which in (('Rw', C.T_RW), ('Re', C.T_RE), ('W', C.T_W), ('b', C.T_B)):
        out[name] = m._engine.get_tensor(which).copy()
    info = m.comm_info()
    out['exchange'] = np.str_(info['exchange_kind'] if info else 'none')
    out['transport'] = np.str_(info['transport'] if info else 'none')
    out['rccl_ranks'] = np.int64(info['rccl_ranks'] if info else 0)
    out['rccl_lib'] = np.str_(loaded_rccl())
    return out


def run(kind):
    """Train two epochs + evaluate; identical code for any world size (the model's
    batch_size is the GLOBAL batch).  Returns a dict of numpy results."""
    from sert_amd import models
    # (the sampler seed is a CLASS attribute: run_c2 / run_soak pin one; a test process that runs several kinds one
    #  after the other must not carry it over -- the ranks of the other kinds start fresh and draw theirs from np.random)
    saved = models.VectorSpaceLanguageModel.sampler_seed
    try:
        if kind == 'c2':
            return run_c2(kind)
        if kind == 'soak':
            return run_soak()
        models.VectorSpaceLanguageModel.sampler_seed = None
        return _run_epochs(kind)
    finally:
        models.VectorSpaceLanguageModel.sampler_seed = saved


def _run_epochs(kind):
    from sert_amd import models
    from tests import util as U
    B, n, z, Vw, Ve, d = 96, 3, 4, 200, 20, 16          # 96 rows: 2, 3 and 4 ranks
    if kind == 'loglinear_bigw':
        Ve, d = 70000, 64                               # dense W: 4.48 M elements (> 2^22: a sharded tensor), no R_e
    if kind == 'vectorspace':
        p = U.make_vs_problem(61, B * 6 + 5, n, z, Vw, Ve, d, d)      # +5: an incomplete tail
        pv = U.make_vs_problem(62, B * 2, n, z, Vw, Ve, d, d)
        np.random.seed(5)
        m = models.VectorSpaceLanguageModel(
            batch_size=B, window_size=n, num_negative_samples=z, representations_init=p['Rw'],
            entity_representations_init=p['Re'], regularization_lambda=0.01,
            training_set=(p['X'], p['y'], p['w']), validation_set=(pv['X'], pv['y']))
    else:
        p = U.make_ll_problem(61, B * 6 + 5, n, Vw, Ve, d, 'int')
        pv = U.make_ll_problem(62, B * 2, n, Vw, Ve, d, 'int')
        np.random.seed(5)
        m = models.LanguageModel(
            batch_size=B, window_size=n, representations_init=p['Rw'], output_layer_size=Ve,
            regularization_lambda=0.01, training_set=(p['X'], p['y'], p['w']),
            validation_set=(pv['X'], pv['y']))
    out = {}
    np.random.seed(7)
    out['epoch1'] = np.float64(m.train()[1])
    out['epoch2'] = np.float64(m.train()[1])
    out['train_error'] = np.float64(m.train_error()[0])
    out['validation_error'] = np.float64(m.validation_error()[0])
    from sert_amd import _capi as C
    for name, which in (('Rw', C.T_RW), ('W', C.T_W), ('b', C.T_B)) + \
            ((('Re', C.T_RE),) if kind == 'vectorspace' else ()):
        out[name] = m._engine.get_tensor(which).copy()
    # the optimiser state (data parallel: sharded over the ranks, gathered by this collective read)
    st = m.get_optimizer_state()
    out['step'] = np.int64(st.pop('step'))
    for name, value in st.items():
        out['opt_' + name] = value
    info = m.comm_info()
    out['exchange'] = np.str_(info['exchange_kind'] if info else 'none')
    out['comm_world'] = np.int64(m._ctx.world_size)
    out['comm_bytes_per_step'] = np.float64(info['comm_bytes_per_step'] if info else 0.0)
    out['transport'] = np.str_(info['transport'] if info else 'none')
    out['rccl_ranks'] = np.int64(info['rccl_ranks'] if info else 0)
    out['rccl_lib'] = np.str_(loaded_rccl())
    return out


def main():
    kind, path = sys.argv[1], sys.argv[2]
    from sert_amd import distributed as dist, models
    ctx = dist.init_from_env()
    models.ModelBase.device = 0          # every rank on the only GPU
    out = run(kind)
    if ctx.rank == 0:
        np.savez(path, **out)
    dist.barrier()
    dist.shutdown()


if __name__ == '__main__':
    main()
