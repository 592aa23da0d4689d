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


def run(kind):
    """Train two epochs + evaluate; identical code for any world size (the model's
    batch_size is the GLOBAL batch).  Returns a dict of numpy results."""
    from sert_amd import models
    from tests import util as U
    B, n, z, Vw, Ve, d = 64, 3, 4, 200, 20, 16
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
