"""Epoch driver and data-set plumbing behind ``bin/train.py``.

Behavioural contract (what the reference's bin/train.py does, lines cited; the
code here is organised differently):

  * data.npz -> (x, y, w) tuples; missing / ignored ``w_train`` means unit
    weights (train.py:79-90);
  * ``--one_hot_classes``: one (instance, entity) pair per non-zero of the sparse
    truth matrix, rows of x / w repeated accordingly (train.py:103-119, :186-245);
  * the driver measures train + validation error BEFORE any training, dumps
    ``<out>_0.bin``, then per epoch: train, measure, dump ``<out>_<epoch>.bin``;
    it stops early when the training error moves by less than the abort
    threshold (train.py:262-348);
  * a dump is a stream of pickles: the extra objects (the CLI namespace) followed
    by ``model.get_state()`` (train.py:289-300).
"""
import logging
import os
import pickle

import numpy as np
import scipy.sparse

from sert_amd import distributed, models


def error_delta(error):
    """(absolute, relative) change between the last two entries; (0, 0) while
    fewer than two measurements exist."""
    if len(error) < 2:
        return 0.0, 0.0
    previous, current = error[-2], error[-1]
    change = current - previous
    return change, change / float(previous)


def sparse_to_one_hot_multiple(y, *matrices):
    """Expand a sparse (N, V_e) truth matrix into one int32 entity id per stored
    non-zero, in row-major order, and repeat the matching row of every extra
    dense matrix once per non-zero.  Returns ``(ids, [expanded matrices])``.
    Every instance must own at least one non-zero (RuntimeError otherwise)."""
    assert scipy.sparse.issparse(y), 'Matrix y should be sparse.'
    num_instances, num_classes = y.shape
    assert num_classes < (1 << 31), \
        'Number of classes should be encodable in 32-bit signed integer.'
    for matrix in matrices:
        assert isinstance(matrix, np.ndarray), \
            'Matrix {0} should be dense.'.format(repr(matrix))
        assert matrix.shape[0] == num_instances

    triplets = y.tocoo()
    owner, entity = triplets.row, triplets.col
    if owner.size:
        # owners must read 0,0..,1,1..,2,..: a jump means an instance without truth
        jumps = np.diff(owner, prepend=-1)
        if ((jumps < 0) | (jumps > 1)).any():
            raise RuntimeError(
                'Every truth value should have at least one non-zero index.')

    ids = entity.astype(np.int32)
    expanded = [np.ascontiguousarray(matrix[owner]).astype(matrix.dtype, copy=False)
                for matrix in matrices]
    return ids, expanded


def load_data_sets(path, ignore_weights=False):
    """-> (training_set, validation_set) from a prepare.py-style npz."""
    logging.info('Loading data from %s.', path)
    archive = np.load(path, allow_pickle=True)
    x_train = archive['x_train']
    if 'w_train' in archive and not ignore_weights:
        weights = archive['w_train']
    else:
        logging.warning('No weights found in data set; '
                        'assuming uniform instance weighting.')
        weights = np.ones(x_train.shape[0], dtype=np.float32)
    # the truth matrices are 0-d object arrays wrapping a csr_matrix
    training_set = (x_train, archive['y_train'][()], weights)
    validation_set = (archive['x_validate'], archive['y_validate'][()])
    for name, parts in (('Training', training_set), ('Validation', validation_set)):
        logging.info('%s instances: %s', name,
                     ' '.join('%s (%s)' % (p.shape, p.dtype) for p in parts))
    return training_set, validation_set


def to_one_hot(training_set, validation_set):
    """--one_hot_classes applied to both splits."""
    logging.info('Transforming y-values to one-hot values.')
    if not (scipy.sparse.issparse(training_set[1]) and
            scipy.sparse.issparse(validation_set[1])):
        raise RuntimeError(
            'Argument --one_hot_classes expects sparse truth values.')
    y_train, (x_train, w_train) = sparse_to_one_hot_multiple(
        training_set[1], training_set[0], training_set[2])
    y_validate, (x_validate,) = sparse_to_one_hot_multiple(
        validation_set[1], validation_set[0])
    return (x_train, y_train, w_train), (x_validate, y_validate)


class EpochDriver(object):
    """State of one training run: error histories and the dump routine."""

    def __init__(self, model, output_path, extra_objects=(),
                 save_optimizer_state=False):
        assert isinstance(model, models.ModelInterface)
        self.model = model
        self.output_path = output_path
        self.extra_objects = list(extra_objects)
        self.save_optimizer_state = save_optimizer_state
        self.means = {'training': [], 'validation': []}
        self.stddevs = {'training': [], 'validation': []}
        self.writes_files = distributed.get_context().rank == 0

    def measure(self):
        for split, fn in (('training', self.model.train_error),
                          ('validation', self.model.validation_error)):
            mean, stddev = fn()
            self.means[split].append(mean)
            self.stddevs[split].append(stddev)

    def dump(self, epoch):
        state = list(self.model.get_state())      # every rank reads its replica
        trailer = None
        if self.save_optimizer_state and hasattr(self.model, 'get_optimizer_state'):
            # (data parallel: the optimiser state is sharded over the ranks -- gathering it is
            # a collective, so every rank takes part even though only rank 0 writes)
            trailer = {'optimizer_state': self.model.get_optimizer_state(),
                       'sampler_state': self.model.get_sampler_state(),
                       'numpy_random_state': np.random.get_state(),
                       'epoch': epoch,
                       'errors': {'means': self.means, 'stddevs': self.stddevs}}
        if not self.writes_files:
            return
        filename = '{0}_{1}.bin'.format(self.output_path, epoch)
        with open(filename, 'wb') as f:
            for obj in self.extra_objects + state:
                pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
            if trailer is not None:
                # trailing pickle; readers that stop after the representations ignore it
                pickle.dump(trailer, f, protocol=pickle.HIGHEST_PROTOCOL)
        logging.info('Saved model "%s" (%d megabyte).', filename,
                     os.path.getsize(filename) / 1024 / 1024)

    def report(self):
        for split, label in (('training', 'Training'), ('validation', 'Validation')):
            logging.info('%s errors: %s; delta=%s', label,
                         list(zip(self.means[split], self.stddevs[split])),
                         error_delta(self.means[split]))

    def stalled(self, threshold):
        history = self.means['training']
        return len(history) > 1 and abs(history[-1] - history[-2]) < threshold

    def validation_got_worse(self):
        history = self.means['validation']
        return history[-1] > history[-2]


def read_checkpoint(path):
    """A dump written by EpochDriver.dump (or by the reference's bin/train.py:289-300 layout):
    -> dict(args, predict_fn, tables=[R_w(, R_e)], trailer or None)."""
    objects = []
    with open(path, 'rb') as f:
        while True:
            try:
                objects.append(pickle.load(f))
            except EOFError:
                break
    if len(objects) < 3:
        raise RuntimeError('%s does not look like a model dump (%d objects).' % (path, len(objects)))
    trailer = None
    if isinstance(objects[-1], dict) and 'optimizer_state' in objects[-1]:
        trailer = objects.pop()
    return {'args': objects[0], 'predict_fn': objects[1], 'tables': objects[2:], 'trailer': trailer}


def restore(model, checkpoint):
    """Put a model (freshly constructed on the same data, with the checkpoint's sampler seed)
    into the state the dump was taken in: parameters, optimiser tensors and step, sampler
    positions, the global numpy generator (the next epoch's batch shuffle draws from it).
    Returns the epoch the dump belongs to, or None when it carries no resume trailer."""
    from sert_amd import _capi
    tables, fn = checkpoint['tables'], checkpoint['predict_fn']
    state = fn.__getstate__()
    model._engine.set_tensor(_capi.T_RW, tables[0])
    if len(tables) > 1:
        model._engine.set_tensor(_capi.T_RE, tables[1])
    model.set_dense(state['W'], state['b'])
    trailer = checkpoint['trailer']
    if trailer is None:
        logging.warning('Checkpoint holds no optimiser state (written without '
                        '--save_optimizer_state): the optimiser restarts from zero moments.')
        return None
    model.set_optimizer_state(trailer['optimizer_state'])
    model.set_sampler_state(trailer['sampler_state'])
    np.random.set_state(trailer['numpy_random_state'])
    return int(trailer['epoch'])


def train(model, num_epochs, output_path,
          abort_threshold=1e-5, early_stopping=False,
          additional_args=[], save_optimizer_state=False, resume_from=None):
    """The reference's epoch driver (train.py:262-348).  ``resume_from`` (additive): a
    checkpoint from read_checkpoint already applied with restore(); the driver then continues
    behind the epoch it was written at instead of starting with measure + dump 0."""
    assert isinstance(abort_threshold, float)
    run = EpochDriver(model, output_path, additional_args, save_optimizer_state)

    first_epoch = 1
    if resume_from is not None and resume_from.get('trailer') is not None:
        trailer = resume_from['trailer']
        run.means = {k: list(v) for k, v in trailer['errors']['means'].items()}
        run.stddevs = {k: list(v) for k, v in trailer['errors']['stddevs'].items()}
        first_epoch = int(trailer['epoch']) + 1
        logging.info('Resuming behind epoch %d.', first_epoch - 1)
    else:
        run.measure()
        run.dump(0)

    for epoch in range(first_epoch, num_epochs + 1):
        logging.info('Epoch %d.', epoch)
        num_batches, mean_cost = model.train()
        logging.info('Epoch %d: processed %d batches; average error=%f.',
                     epoch, num_batches, mean_cost)

        logging.info('Epoch %d: measuring training/validation error.', epoch)
        run.measure()
        run.report()
        run.dump(epoch)

        assert np.all(np.isfinite(run.means['training'][-1]))

        if early_stopping:
            assert np.all(np.isfinite(run.means['validation'][-1]))
            if run.validation_got_worse():
                logging.info('Validation error stopped decreasing; aborting.')
                return

        if run.stalled(abort_threshold):
            logging.error('No learning was performed during '
                          'the last iteration; aborting.')
            return
