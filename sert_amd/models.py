"""Drop-in for ``sert.models`` backed by libsert_hip.so (MI355X / gfx950).

Mirrors the reference's class surface (sert/models.py): same class names, same
constructor keyword arguments, same methods and return values, same exceptions
-- but the four compiled Theano functions (train_fn / test_fn / validate_fn /
predict_fn, models.py:581-636) are calls into the HIP engine through the C ABI
declared in include/sert_hip.h.  There is no CPU execution path: constructing a
model without a visible MI355X raises.

Differences that are deliberate and additive:
  * ``get_representations()`` returns fresh host copies of the device tensors
    (the reference returns borrowed views of Theano shared variables,
    models.py:183, :945);
  * ``predict_fn`` is a small picklable callable carrying (W, b[, R_w]) instead
    of a pickled compiled Theano function (models.py:670-680);
  * ``get_optimizer_state()/set_optimizer_state()`` expose the optimiser state
    for checkpoint/resume (the reference cannot resume);
  * negatives are drawn on the device (Philox) unless ``negative_sampler`` is
    set to a callable ``batch_index -> (B, z) int64`` (parity runs).
"""
import logging
import os
import time

import numpy as np
import scipy.sparse as sparse

from sert_amd import _capi
from sert_amd import distributed


class _EpochPass(object):
    """One pass over the complete batches of a split: which batches, in which order, the
    per-result finite check and the throughput line.  Shared by every way this module walks
    a split -- one host synchronisation per batch (the reference's loop, models.py:351-399)
    or one per chunk of batches (train() with steps_per_sync > 1, the error passes).

    Contract kept from the reference: N // B batches and a warning about a dropped tail;
    training shuffles the batch ORDER with the global np.random; a non-finite result raises
    RuntimeError naming the position; "batches per second" is logged every
    ``report_interval`` results and at the end (pairs/s = that x batch_size)."""

    def __init__(self, batch_size, num_instances, report_interval, shuffle):
        self.started = time.time()
        self.total = num_instances // batch_size
        self.report_interval = report_interval
        leftover = num_instances - self.total * batch_size
        if leftover > 0:
            logging.warning('\tIgnoring incomplete batch of size %d.', leftover)
        self.order = list(range(self.total))
        if shuffle:
            logging.debug('Shuffling batches.')
            np.random.shuffle(self.order)
        self.results = []

    def successor(self, position):
        """Batch that follows the one at `position`, or None at the end of the pass."""
        return self.order[position + 1] if position + 1 < self.total else None

    def accept(self, value):
        self.results.append(value)
        done = len(self.results)
        if not np.all(np.isfinite(value)):
            raise RuntimeError(
                'Encountered NaN or infinity ({error}) during batch iteration '
                '(batch {batches_finished}/{num_batches}).'.format(
                    error=value, batches_finished=done, num_batches=self.total))
        if done % self.report_interval == 0 or done == self.total:
            self._report(done)

    def _report(self, done):
        rate = done / max(float(time.time() - self.started), 1e-9)
        eta = (self.total - done) / rate
        logging.info('\tProcessed %d batches; %.2f batches per second; '
                     '%d minutes %d seconds remaining.', done, rate, eta / 60, eta % 60)

    def outcome(self):
        return self.total, self.results


class ModelInterface(object):
    """sert/models.py:295-411: the abstract model the drivers talk to."""

    __DETECT_EXCEPTIONS__ = False

    TRAIN, VALIDATE, TEST = range(2, 5)

    def __init__(self, batch_size):
        assert batch_size > 0
        self.batch_size = batch_size
        logging.debug('Batch size: %d', self.batch_size)

    @classmethod
    def _get_batch_slice(cls, batch_index, batch_size):
        return slice(batch_index * batch_size, batch_index * batch_size + batch_size)

    def _number_of_batches(self, num_instances):
        return num_instances // self.batch_size

    def _iterate_batches(self, fn, num_instances,
                         report_interval=10000, shuffle=False):
        """fn(batch_index) over one pass, one call (and one host synchronisation) per batch."""
        walk = _EpochPass(self.batch_size, num_instances, report_interval, shuffle)
        # The order is known up front: while training, tell the engine which batch follows,
        # so that it can enqueue that batch's work before the host starts waiting for the
        # current loss (no effect on results).
        announce = None
        if fn == getattr(self, 'train_fn', None):
            announce = getattr(getattr(self, '_engine', None), 'hint_next_batch', None)
        for position, batch_index in enumerate(walk.order):
            if announce is not None:
                announce(walk.successor(position))
            walk.accept(fn(batch_index))
        return walk.outcome()

    def _iterate_chunks(self, issue, num_instances, chunk, report_interval, shuffle):
        """issue(list of batch indices) -> their results, one host synchronisation per chunk;
        same order, results, log lines and exception as _iterate_batches (a non-finite result
        surfaces at the end of the chunk that holds it)."""
        walk = _EpochPass(self.batch_size, num_instances, report_interval, shuffle)
        for first in range(0, walk.total, chunk):
            for value in issue(walk.order[first:first + chunk]):
                walk.accept(value)
        return walk.outcome()

    # the four methods every model supplies (sert/models.py:403-413)
    def _not_here(self, *unused_args, **unused_kwargs):
        raise NotImplementedError()

    train = train_error = validation_error = get_state = _not_here


def _as_id_array(x):
    """Token ids as uint8/16/32 (np.min_scalar_type output, prepare.py:380)."""
    x = np.asarray(x)
    if x.dtype in (np.uint8, np.uint16, np.uint32):
        return np.ascontiguousarray(x)
    assert x.size == 0 or (x.min() >= 0 and x.max() < (1 << 32)), \
        'token ids must fit an unsigned 32-bit integer'
    return np.ascontiguousarray(x, dtype=np.uint32)


class ModelBase(ModelInterface):
    """sert/models.py:414-683.  ``learning_method`` is 'adam' or 'adadelta'
    (the reference passes lasagne.updates.adam / adadelta)."""

    __DEBUG__ = False

    #: device ordinal override; None = LOCAL_RANK when data-parallel, else 0
    device = None
    #: callable batch_index -> (B, z) int64 negatives, or None (device sampler)
    negative_sampler = None
    #: same, for train_error()/validation_error(); None = device sampler
    eval_negative_sampler = None
    #: seed of the device sampler (the reference seeds RandomStreams from the
    #: global np.random, models.py:958-959)
    sampler_seed = None
    #: additive throughput knob for train(): number of consecutive batches issued
    #: to the device before their losses are read back.  1 (default) = the
    #: reference's behaviour (one host synchronisation and finite-check per batch,
    #: models.py:369-379); k > 1 defers the check by up to k-1 batches.
    steps_per_sync = 1
    #: batches per host synchronisation in train_error() / validation_error().  Their
    #: per-batch results are only averaged (models.py:649-668), so the passes issue
    #: chunks of batches and check for non-finite values per chunk; 1 = one
    #: synchronisation per batch as in the reference loop.  Same values either way.
    eval_batches_per_sync = 32

    def __init__(self, batch_size,
                 training_set, validation_set,
                 learning_method):
        super(ModelBase, self).__init__(batch_size)
        self.learning_method = learning_method

        x_train, y_train = training_set[0], training_set[1]
        x_validate, y_validate = validation_set[0], validation_set[1]
        self.training_num_instances = y_train.shape[0]
        self.validation_num_instances = y_validate.shape[0]

        # instances are rows of `num_instance_features` ids (models.py:437-446)
        assert x_train.ndim == 2
        self.num_instance_features = int(x_train.shape[1])
        if x_validate.size:
            assert x_validate.shape[1] == self.num_instance_features
        else:
            # an empty validation split may arrive with any shape (models.py:448-454)
            x_validate = x_validate.reshape(0, self.num_instance_features)
            validation_set = (x_validate,) + tuple(validation_set[1:])

        logging.info('Data set contains %d training instances '
                     'and %d validation instances',
                     self.training_num_instances, self.validation_num_instances)

        assert x_train.dtype == x_validate.dtype
        assert y_train.dtype == y_validate.dtype
        self.input_dtype, self.output_dtype = x_train.dtype, y_train.dtype
        self.training_set, self.validation_set = training_set, validation_set

        self._engine = None

    # -- engine plumbing ------------------------------------------------------
    def _create_engine(self, kind, vocab_size, word_dim, num_entities,
                       entity_dim, num_negatives, optimizer):
        """Replaces _create_functions (models.py:530-628): allocate the device
        model and upload the whole data set once (models.py:470-480)."""
        _capi.require_gpu()

        sparse_truths = [sparse.isspmatrix_csr(split[1])
                         for split in (self.training_set, self.validation_set)]
        if sparse_truths[0] != sparse_truths[1]:
            raise RuntimeError('Either training or validation truths are '
                               'sparse while the other is dense.')
        is_training_y_sparse = sparse_truths[0]

        ctx = distributed.get_context()
        if self.batch_size % ctx.world_size != 0:
            raise RuntimeError(
                'batch_size (%d) must be divisible by the number of ranks (%d).'
                % (self.batch_size, ctx.world_size))
        self._ctx = ctx
        self._local_batch = self.batch_size // ctx.world_size

        x_train = _as_id_array(self.training_set[0])
        x_val = _as_id_array(self.validation_set[0])
        if x_val.dtype != x_train.dtype:
            x_val = x_val.astype(x_train.dtype)

        device = self.device
        if device is None and os.environ.get('SERT_DEVICE'):
            device = int(os.environ['SERT_DEVICE'])   # e.g. several ranks on one GPU (tests)
        if device is None:
            device = ctx.local_rank if ctx.world_size > 1 else 0

        seed = self.sampler_seed
        if seed is None:
            seed = int(np.random.randint(low=0, high=(1 << 30)))  # models.py:958-959
        seed = distributed.broadcast_object(seed)  # one sampler stream for all ranks

        opt = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8) \
            if optimizer == 'adam' else \
            dict(lr=1.0, beta1=0.95, beta2=0.0, eps=1e-6)

        self._engine = _capi.Engine(
            kind=kind, batch_size=self._local_batch,
            global_batch_size=self.batch_size,
            window_size=self.window_size, vocab_size=vocab_size,
            num_entities=num_entities, word_dim=word_dim,
            entity_dim=entity_dim, num_negatives=num_negatives or 0,
            id_bytes=x_train.dtype.itemsize, device=device,
            keep_grads=1 if self.__DEBUG__ else 0, deterministic=1,
            inference_only=0, lambda_=float(self.regularization_lambda),
            seed=seed, **opt)

        # SERT_FORCE_COMM=1: run the exchange path (RCCL, world of one) on a single
        # GPU -- for profiling the data-parallel step structure on a 1-GPU box
        if ctx.world_size > 1 and os.environ.get('SERT_COMM') == 'host':
            # verification transport: the exchange goes through pinned host memory and
            # gloo, so that several ranks can share one GPU (tests on a 1-GPU box)
            self._engine.comm_init_host(ctx.rank, ctx.world_size,
                                        distributed.host_alltoall)
        elif ctx.world_size > 1 or os.environ.get('SERT_FORCE_COMM') == '1':
            self._engine.comm_init(ctx.unique_id(), ctx.rank, ctx.world_size)

        self._upload(_capi.SPLIT_TRAIN, x_train, self.training_set[1],
                     self.training_set[2])
        self._upload(_capi.SPLIT_VALIDATE, x_val, self.validation_set[1], None)

    def comm_info(self):
        """Additive: what the data-parallel exchange of this model moves (None when single-process)."""
        if self._engine is None or self._ctx.world_size <= 1:
            return None
        st = self._engine.comm_stats()
        names = {'rows': 'word table owned by rows: all-to-all of the touched parameter rows before the forward and '
                         'of their gradient rows after the backward (static per-batch lists), dense optimiser on the '
                         'owned rows',
                 'zero1': 'ZeRO-1 word table (reduce-scatter + sharded optimiser + all-gather)', 'none': 'replicated'}
        return {'exchange': names[st['exchange']], 'exchange_kind': st['exchange'],
                'rccl_ranks': st['world'] if st['transport'] == 'rccl' else 0, 'transport': st['transport'],
                'comm_bytes_per_step': st['bytes_per_step'], 'zero1_comm_bytes_per_step': st['zero1_bytes_per_step'],
                'steps_counted': st['steps'], 'rows_fetched_per_batch': st['rows_fetched_per_batch'],
                'rows_served_per_batch': st['rows_served_per_batch'],
                'note': 'bytes this rank sends + receives per training step through collectives (word-table exchange + '
                        'the small all-reduce), mean over the steps run so far'}

    def _shard_rows(self, num_instances):
        """Row indices this rank owns (None when single-process)."""
        ctx = self._ctx
        if ctx.world_size == 1:
            return None
        return distributed.shard_rows(num_instances, self.batch_size,
                                      ctx.rank, ctx.world_size)

    def _upload(self, split, x, y, w):
        rows = self._shard_rows(x.shape[0])
        if rows is not None:
            x = x[rows]
            y = y[rows]
            w = w[rows] if w is not None else None
        if sparse.issparse(y):
            y = sparse.csr_matrix(y, dtype=np.float32)
            y.sum_duplicates()
            self._engine.upload_dataset(split, x, csr=y, w=w)
        elif np.asarray(y).ndim == 1:
            self._engine.upload_dataset(split, x, y_int=np.asarray(y), w=w)
        else:
            # dense (N, V_e) truth matrix (T.fmatrix('y'), models.py:735-737)
            self._engine.upload_dataset(
                split, x, csr=sparse.csr_matrix(np.asarray(y, dtype=np.float32)),
                w=w)

    def _local_negatives(self, sampler, batch_index):
        """Explicit negatives of one batch as the engine wants them: this rank's rows of the
        (global_batch, z) array a sampler returns (or the local block, if it already is one)."""
        if sampler is None:
            return None
        neg = np.asarray(sampler(batch_index))
        z = self._engine.cfg.num_negatives
        if self._ctx.world_size > 1 and neg.shape == (self.batch_size, z):
            first = self._ctx.rank * self._local_batch
            neg = neg[first:first + self._local_batch]
        assert neg.shape == (self._local_batch, z), \
            'negative sampler returned %r, expected (%d, %d)' % (neg.shape, self._local_batch, z)
        return neg

    # -- the three step functions (models.py:581-608) ---------------------------
    def train_fn(self, batch_index):
        return self._engine.train_batch(
            batch_index, self._local_negatives(self.negative_sampler, batch_index))

    def test_fn(self, batch_index):
        return self._engine.eval_batch(
            _capi.SPLIT_TRAIN, batch_index,
            self._local_negatives(self.eval_negative_sampler, batch_index))

    def validate_fn(self, batch_index):
        return self._engine.eval_batch(
            _capi.SPLIT_VALIDATE, batch_index,
            self._local_negatives(self.eval_negative_sampler, batch_index))

    # -- passes over a split (models.py:638-668) ----------------------------------
    def train(self):
        count = self.training_num_instances
        logging.info('Training on %d training instances (%d batches).',
                     count, self._number_of_batches(count))
        with _capi.profile_range('sert train pass'):      # (a roctx range; a no-op without the tools library)
            if self.steps_per_sync > 1 and self.negative_sampler is None:
                num_batches, losses = self._iterate_chunks(
                    self._engine.train_batches, count, int(self.steps_per_sync),
                    report_interval=1000, shuffle=True)
            else:
                num_batches, losses = self._iterate_batches(
                    self.train_fn, count, report_interval=1000, shuffle=True)
        return num_batches, np.mean(losses)

    def _error_pass(self, fn, split, count):
        """(mean, std) of the per-batch evaluation losses of a split."""
        chunk = int(self.eval_batches_per_sync)
        with _capi.profile_range('sert error pass'):
            if chunk > 1 and self.eval_negative_sampler is None:
                _, losses = self._iterate_chunks(
                    lambda batches: self._engine.eval_batches(
                        split, np.asarray(batches, dtype=np.int64)),
                    count, chunk, report_interval=10000, shuffle=False)
            else:
                _, losses = self._iterate_batches(fn, count)
        return np.mean(losses), np.std(losses)

    def train_error(self):
        count = self.training_num_instances
        logging.info('Measuring error on %d training instances (%d batches).',
                     count, self._number_of_batches(count))
        return self._error_pass(self.test_fn, _capi.SPLIT_TRAIN, count)

    def validation_error(self):
        count = self.validation_num_instances
        logging.info('Measuring error on %d validation instances (%d batches).',
                     count, self._number_of_batches(count))
        return self._error_pass(self.validate_fn, _capi.SPLIT_VALIDATE, count)

    def get_state(self):
        """[predict_fn, R_w(, R_e)] -- what bin/train.py pickles behind its namespace
        (models.py:670-680).

        DATA PARALLEL: COLLECTIVE once a training step has run.  With the word table owned by rows
        (the default exchange) a rank's copy of R_w is current only where it owns or has fetched, so
        reading it -- here, get_representations(), predict_fn of the loglinear model, train_error /
        validation_error, get_optimizer_state(), close() -- first all-gathers the owned slabs
        (sert_get_tensor(SERT_T_RW) / sert_eval_batches / sert_predict_tokens / sert_comm_destroy).
        EVERY rank must make the call, in the same order (sert_amd.training does); a call from rank 0
        alone -- a checkpoint hook, a callback -- blocks until the peers make it too (host transport:
        RuntimeError 'rendezvous timed out' after the FileStore deadline; RCCL: no deadline)."""
        tables = self.get_representations()
        if not isinstance(tables, (tuple, list)):
            tables = (tables,)
        return [self.predict_fn] + list(tables)

    def get_representations(self):
        raise RuntimeError()

    # -- additive: optimiser state for checkpoint / resume ---------------------
    _STATE_TENSORS = ()

    def get_sampler_state(self):
        """Seed and evaluation position of the device negative sampler (the training position
        is the optimiser step).  A run resumed with the same seed and positions draws the
        negatives the uninterrupted run would have drawn."""
        return {'seed': int(self._engine.cfg.seed), 'eval_draws': self._engine.get_eval_draws()}

    def set_sampler_state(self, st):
        if int(st['seed']) != int(self._engine.cfg.seed):
            raise RuntimeError('the sampler seed is fixed at construction: set %s.sampler_seed = %d '
                               'before creating the model' % (type(self).__name__, st['seed']))
        self._engine.set_eval_draws(st['eval_draws'])

    def get_optimizer_state(self):
        """Additive.  Data parallel: COLLECTIVE (the sharded moments are gathered) -- every rank calls it."""
        st = {'step': self._engine.get_step()}
        for name, which, shape in self._STATE_TENSORS:
            st[name] = self._engine.get_tensor(which, shape(self))
        return st

    def set_optimizer_state(self, st):
        self._engine.set_step(st['step'])
        for name, which, _ in self._STATE_TENSORS:
            self._engine.set_tensor(which, st[name])


class LanguageModelBase(ModelBase):
    """sert/models.py:686-801."""

    def __init__(self,
                 window_size,
                 representations_init,
                 regularization_lambda,
                 regularization_fn,
                 **kwargs):
        super(LanguageModelBase, self).__init__(**kwargs)
        assert window_size >= 1
        self.window_size = window_size
        self.initial_representations = representations_init
        self.vocabulary_size, self.representation_size = representations_init.shape[:2]
        self.regularization_lambda, self.regularization_fn = \
            regularization_lambda, regularization_fn
        assert self.num_instance_features == self.window_size

    def get_representations(self):
        """Host copy of R_w (models.py:797-801).  Data parallel: COLLECTIVE after a training step --
        every rank must call it (see get_state)."""
        if self._engine is not None:
            return self._engine.get_tensor(
                _capi.T_RW, (self.vocabulary_size, self.representation_size))
        else:
            return None


def _glorot_uniform(shape):
    """lasagne.init.GlorotUniform().sample(shape) from the global np.random
    [upstream Lasagne 0.1]; DenseLayer's W default (models.py:846, :1057)."""
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return np.random.uniform(low=-a, high=a, size=shape).astype(np.float32)


def l2_regularization(objects):
    """Marker kept for signature parity (models.py:92-120); the L2 term is
    computed inside the fused optimiser kernel."""
    raise NotImplementedError('l2 is evaluated on the device')


class LogLinearPredictFn(object):
    """Picklable stand-in for the loglinear predict_fn (models.py:880-890):
    f(batch[B,n] uint, mask[B,n] int8) -> (B,n,V_e) float32 per-token
    distributions.  The mask is ignored, as in the reference
    (on_unused_input='warn')."""

    def __init__(self, model=None, R_w=None, W=None, b=None, window_size=None):
        self._model = model
        self.R_w, self.W, self.b = R_w, W, b
        self.window_size = window_size
        self._engine = None

    def __getstate__(self):
        if self._model is not None:
            m = self._model
            return dict(R_w=m.get_representations(), W=m.get_dense_weights(),
                        b=m.get_dense_bias(), window_size=m.window_size)
        return dict(R_w=self.R_w, W=self.W, b=self.b,
                    window_size=self.window_size)

    def __setstate__(self, st):
        self._model = None
        self._engine = None
        self.R_w, self.W, self.b = st['R_w'], st['W'], st['b']
        self.window_size = st['window_size']

    def _get_engine(self, id_bytes):
        if self._model is not None:
            return self._model._engine
        if self._engine is None or self._engine.cfg.id_bytes != id_bytes:
            _capi.require_gpu()
            Vw, d = self.R_w.shape
            self._engine = _capi.Engine(
                kind=_capi.KIND_LOGLINEAR, batch_size=1, global_batch_size=1,
                window_size=self.window_size, vocab_size=Vw,
                num_entities=self.W.shape[1], word_dim=d, entity_dim=0,
                num_negatives=0, id_bytes=id_bytes, device=0, keep_grads=0,
                deterministic=1, inference_only=1, lambda_=0.0,
                lr=1.0, beta1=0.95, beta2=0.0, eps=1e-6, seed=0)
            self._engine.set_tensor(_capi.T_RW, self.R_w)
            self._engine.set_tensor(_capi.T_W, self.W)
            self._engine.set_tensor(_capi.T_B, self.b)
        return self._engine

    def __call__(self, batch, mask=None):
        batch = _as_id_array(batch)
        assert batch.ndim == 2
        eng = self._get_engine(batch.dtype.itemsize)
        if batch.dtype.itemsize != eng.cfg.id_bytes:
            batch = batch.astype({1: np.uint8, 2: np.uint16, 4: np.uint32}[
                eng.cfg.id_bytes])
        return eng.predict_tokens(batch)


class LanguageModel(LanguageModelBase):
    """'loglinear' (sert/models.py:804-890): per-token softmax over all
    entities -> log-product over the window -> renormalise -> clipped
    categorical cross-entropy; Adadelta."""

    _STATE_TENSORS = (
        ('accu_R_w', _capi.T_STATE0_RW, lambda m: (m.vocabulary_size, m.representation_size)),
        ('accu_W', _capi.T_STATE0_W, lambda m: (m.representation_size, m.output_layer_size)),
        ('accu_b', _capi.T_STATE0_B, lambda m: (m.output_layer_size,)),
        ('delta_R_w', _capi.T_STATE1_RW, lambda m: (m.vocabulary_size, m.representation_size)),
        ('delta_W', _capi.T_STATE1_W, lambda m: (m.representation_size, m.output_layer_size)),
        ('delta_b', _capi.T_STATE1_B, lambda m: (m.output_layer_size,)),
    )

    def __init__(self,
                 batch_size, window_size,
                 representations_init,
                 output_layer_size,
                 regularization_lambda,
                 training_set,
                 validation_set):
        super(LanguageModel, self).__init__(
            batch_size=batch_size,
            window_size=window_size,
            representations_init=representations_init,
            regularization_lambda=regularization_lambda,
            regularization_fn=l2_regularization,
            training_set=training_set, validation_set=validation_set,
            learning_method='adadelta')

        self.output_layer_size = output_layer_size

        self._create_engine(
            _capi.KIND_LOGLINEAR, self.vocabulary_size,
            self.representation_size, output_layer_size, 0, 0, 'adadelta')

        # DenseLayer defaults: W GlorotUniform, b zeros (models.py:846-849)
        dense_W = _glorot_uniform((self.representation_size, output_layer_size))
        dense_W = distributed.broadcast_array(dense_W)
        self._engine.set_tensor(_capi.T_RW, representations_init)
        self._engine.set_tensor(_capi.T_W, dense_W)
        self._engine.set_tensor(
            _capi.T_B, np.zeros(output_layer_size, dtype=np.float32))

        self.predict_fn = LogLinearPredictFn(model=self)

    def get_dense_weights(self):
        return self._engine.get_tensor(
            _capi.T_W, (self.representation_size, self.output_layer_size))

    def get_dense_bias(self):
        return self._engine.get_tensor(_capi.T_B, (self.output_layer_size,))

    def set_dense(self, W, b):
        self._engine.set_tensor(_capi.T_W, W)
        self._engine.set_tensor(_capi.T_B, b)


class VectorSpacePredictFn(object):
    """Picklable stand-in for the vectorspace predict_fn (models.py:1107-1118):
    f(avg_word_embedding[d_w]) -> (1, d_e) = tanh(avg.W + b), no clip.  Also
    accepts a (Q, d_w) matrix and then returns (Q, d_e) (batched queries)."""

    def __init__(self, model=None, W=None, b=None):
        self._model = model
        self.W, self.b = W, b
        self._engine = None

    def __getstate__(self):
        if self._model is not None:
            return dict(W=self._model.get_dense_weights(),
                        b=self._model.get_dense_bias())
        return dict(W=self.W, b=self.b)

    def __setstate__(self, st):
        self._model = None
        self._engine = None
        self.W, self.b = st['W'], st['b']

    def _get_engine(self):
        if self._model is not None:
            return self._model._engine
        if self._engine is None:
            _capi.require_gpu()
            dw, de = self.W.shape
            self._engine = _capi.Engine(
                kind=_capi.KIND_VECTORSPACE, batch_size=1, global_batch_size=1,
                window_size=1, vocab_size=1, num_entities=1, word_dim=dw,
                entity_dim=de, num_negatives=0, id_bytes=4, device=0,
                keep_grads=0, deterministic=1, inference_only=1, lambda_=0.0,
                lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, seed=0)
            self._engine.set_tensor(_capi.T_W, self.W)
            self._engine.set_tensor(_capi.T_B, self.b)
        return self._engine

    def __call__(self, avg_word_embedding):
        avg = np.asarray(avg_word_embedding, dtype=np.float32)
        eng = self._get_engine()
        return eng.predict_project(avg.reshape(-1, avg.shape[-1]))


class VectorSpaceLanguageModelBase(LanguageModelBase):
    """sert/models.py:905-1021."""

    def __init__(self,
                 batch_size, window_size,
                 num_negative_samples,
                 representations_init,
                 entity_representations_init,
                 regularization_lambda,
                 training_set,
                 validation_set):
        super(VectorSpaceLanguageModelBase, self).__init__(
            batch_size=batch_size,
            window_size=window_size,
            representations_init=representations_init,
            regularization_lambda=regularization_lambda,
            regularization_fn=l2_regularization,
            training_set=training_set, validation_set=validation_set,
            learning_method='adam')

        self.num_entities, self.entity_representation_size = \
            entity_representations_init.shape[:2]
        assert self.training_set[1].ndim == 1, 'Only one-hot vectors supported.'
        if num_negative_samples is not None:
            assert num_negative_samples >= 0, \
                'Number of negative samples should be None, zero or positive ' \
                '(currently: {0}).'.format(num_negative_samples)
        self.num_negative_samples = num_negative_samples

    def get_representations(self):
        """Host copies of (R_w, R_e) (models.py:943-945).  Data parallel: COLLECTIVE after a training
        step -- every rank must call it (see ModelBase.get_state)."""
        return (self._engine.get_tensor(
                    _capi.T_RW,
                    (self.vocabulary_size, self.representation_size)),
                self._engine.get_tensor(
                    _capi.T_RE,
                    (self.num_entities, self.entity_representation_size)))


class VectorSpaceLanguageModel(VectorSpaceLanguageModelBase):
    """'vectorspace' / LSE (sert/models.py:1024-1118): window mean-pool ->
    tanh projection (clipped) -> sigmoid NCE against z uniform negatives;
    Adam."""

    _STATE_TENSORS = (
        ('m_R_e', _capi.T_STATE0_RE, lambda m: (m.num_entities, m.entity_representation_size)),
        ('m_R_w', _capi.T_STATE0_RW, lambda m: (m.vocabulary_size, m.representation_size)),
        ('m_W', _capi.T_STATE0_W, lambda m: (m.representation_size, m.entity_representation_size)),
        ('m_b', _capi.T_STATE0_B, lambda m: (m.entity_representation_size,)),
        ('v_R_e', _capi.T_STATE1_RE, lambda m: (m.num_entities, m.entity_representation_size)),
        ('v_R_w', _capi.T_STATE1_RW, lambda m: (m.vocabulary_size, m.representation_size)),
        ('v_W', _capi.T_STATE1_W, lambda m: (m.representation_size, m.entity_representation_size)),
        ('v_b', _capi.T_STATE1_B, lambda m: (m.entity_representation_size,)),
    )

    def __init__(self,
                 batch_size, window_size,
                 num_negative_samples,
                 representations_init,
                 entity_representations_init,
                 regularization_lambda,
                 training_set,
                 validation_set):
        super(VectorSpaceLanguageModel, self).__init__(
            batch_size=batch_size,
            window_size=window_size,
            num_negative_samples=num_negative_samples,
            representations_init=representations_init,
            entity_representations_init=entity_representations_init,
            regularization_lambda=regularization_lambda,
            training_set=training_set,
            validation_set=validation_set)

        # the reference fails inside _negative_sampling for None / 0
        # (models.py:948); keep the requirement explicit
        assert num_negative_samples is not None and num_negative_samples > 0

        self._create_engine(
            _capi.KIND_VECTORSPACE, self.vocabulary_size,
            self.representation_size, self.num_entities,
            self.entity_representation_size, num_negative_samples, 'adam')

        # 'WordProjection' DenseLayer defaults (models.py:1057-1061)
        dense_W = _glorot_uniform(
            (self.representation_size, self.entity_representation_size))
        dense_W = distributed.broadcast_array(dense_W)
        self._engine.set_tensor(_capi.T_RW, representations_init)
        self._engine.set_tensor(_capi.T_RE, entity_representations_init)
        self._engine.set_tensor(_capi.T_W, dense_W)
        self._engine.set_tensor(
            _capi.T_B,
            np.zeros(self.entity_representation_size, dtype=np.float32))

        self.predict_fn = VectorSpacePredictFn(model=self)

    def get_dense_weights(self):
        return self._engine.get_tensor(
            _capi.T_W,
            (self.representation_size, self.entity_representation_size))

    def get_dense_bias(self):
        return self._engine.get_tensor(
            _capi.T_B, (self.entity_representation_size,))

    def set_dense(self, W, b):
        self._engine.set_tensor(_capi.T_W, W)
        self._engine.set_tensor(_capi.T_B, b)


class VectorSpaceSoftmaxLanguageModel(VectorSpaceLanguageModelBase):
    """ADDITIVE model, not in the reference: the vectorspace
    encoder (window mean-pool -> tanh projection -> clip) scored against ALL
    entities -- logits = p . R_e^T, clipped softmax cross-entropy -- instead of
    NCE against z sampled negatives; same dense L2 and Adam.  This is the
    "embed gather + MFMA projection + full softmax" workload BASELINE.json names
    for the LSE config.  predict_fn and get_state() are those of the vectorspace
    model, so bin/query.py ranks with it unchanged."""

    _STATE_TENSORS = VectorSpaceLanguageModel._STATE_TENSORS

    def __init__(self,
                 batch_size, window_size,
                 representations_init,
                 entity_representations_init,
                 regularization_lambda,
                 training_set,
                 validation_set,
                 num_negative_samples=None):
        super(VectorSpaceSoftmaxLanguageModel, self).__init__(
            batch_size=batch_size,
            window_size=window_size,
            num_negative_samples=None,
            representations_init=representations_init,
            entity_representations_init=entity_representations_init,
            regularization_lambda=regularization_lambda,
            training_set=training_set,
            validation_set=validation_set)

        self._create_engine(
            _capi.KIND_VECTORSPACE_SOFTMAX, self.vocabulary_size,
            self.representation_size, self.num_entities,
            self.entity_representation_size, 0, 'adam')

        dense_W = _glorot_uniform(
            (self.representation_size, self.entity_representation_size))
        dense_W = distributed.broadcast_array(dense_W)
        self._engine.set_tensor(_capi.T_RW, representations_init)
        self._engine.set_tensor(_capi.T_RE, entity_representations_init)
        self._engine.set_tensor(_capi.T_W, dense_W)
        self._engine.set_tensor(
            _capi.T_B,
            np.zeros(self.entity_representation_size, dtype=np.float32))

        self.predict_fn = VectorSpacePredictFn(model=self)

    get_dense_weights = VectorSpaceLanguageModel.get_dense_weights
    get_dense_bias = VectorSpaceLanguageModel.get_dense_bias
    set_dense = VectorSpaceLanguageModel.set_dense
