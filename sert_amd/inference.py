"""Drop-in for ``sert.inference`` (query batching in front of predict_fn).

Same entry point and classes as the reference (sert/inference.py):
``create(predict_fn, word_representations, batch_size, window_size,
vocabulary_size, result_callback)`` returns an object with
``submit(query_tokens, **kwargs)`` and ``process()``.

  * loglinear  -> WordBatcher: packs query tokens row-major into a fixed
    (batch_size, window_size) id batch + int8 mask, flushing through
    predict_fn when full (inference.py:28-143);
  * vectorspace -> EmbeddingMapper: mean of the query's word vectors ->
    predict_fn -> callback, one query at a time (inference.py:146-167).

Additive: ``create(..., batched=True)`` returns a BatchedEmbeddingMapper that
defers all queries to ``process()`` and pushes them through predict_fn and the
callback's ``process_batch`` in ONE device call each (the 10k x 100k scoring
config of BASELINE.json).
"""
import logging

import numpy as np


def create(predict_fn, word_representations,
           batch_size, window_size, vocabulary_size,
           result_callback, batched=False):
    """The front end that fits the callback: callbacks that want the pooled query vector (vectorspace) get a mapper,
    the others (loglinear: per-token distributions) the fixed-shape batcher (sert/inference.py:5-25)."""
    assert result_callback is not None
    if not result_callback.should_average_input():
        ids = np.min_scalar_type(vocabulary_size - 1)          # id width of the packed batches (inference.py:10)
        logging.info('Instance elements will be stored using %s.', ids)
        return WordBatcher(predict_fn, batch_size, window_size, ids, result_callback)
    mapper = BatchedEmbeddingMapper if batched and hasattr(result_callback, 'process_batch') else EmbeddingMapper
    return mapper(predict_fn, word_representations, result_callback)


class WordBatcher(object):
    """Queries in front of the loglinear predict_fn, which takes exactly (batch_size, window_size) ids + an int8 mask
    (models.py:830-856).  Same surface and same batches as the reference's batcher (sert/inference.py:28-143, pinned by
    tests/test_golden_host.py), built differently: ``submit`` only QUEUES a query and books its rows; the id batch and the
    mask are filled for all queued queries at once when the batch goes out -- a query of T tokens that starts at row r
    owns the flat slots [r n, r n + T) of the row-major batch (that is what "long queries spill over several rows" means,
    inference.py:115-117, 132-143), so packing is one scatter of the concatenated tokens and a query's per-token
    distributions are one contiguous slice of the flattened result (inference.py:89-94)."""

    OVERFLOW, TRUNCATE = range(5, 7)

    def __init__(self, predict_fn, batch_size, window_size, instance_dtype, result_callback=None, overflow_mode=OVERFLOW):
        assert overflow_mode in (WordBatcher.OVERFLOW, WordBatcher.TRUNCATE)
        assert result_callback is None or callable(result_callback)
        self.predict_fn, self.callback = predict_fn, result_callback
        self.batch_size, self.window_size, self.overflow_mode = batch_size, window_size, overflow_mode
        self.batch = np.zeros((batch_size, window_size), dtype=instance_dtype)     # (padding slots carry token id 0)
        self.mask = np.zeros((batch_size, window_size), dtype=np.int8)
        self._queue = []            # (first row, tokens, kwargs) of the queries waiting for the next predict_fn call
        self._rows = 0              # rows booked by them

    @property
    def num_used_instances(self):
        return self._rows

    def submit(self, query_tokens, **kwargs):
        assert len(query_tokens) > 0
        if self.overflow_mode == WordBatcher.TRUNCATE and len(query_tokens) > self.window_size:
            logging.error('Truncated query "%s" as it exceeded the window size.', query_tokens)
            query_tokens = query_tokens[:self.window_size]
        rows = -(-len(query_tokens) // self.window_size)
        if rows > self.batch_size:
            raise RuntimeError()        # (more tokens than one batch holds: inference.py:124-125)
        if self._rows + rows > self.batch_size:
            self.process()              # (no room left: the batch goes out first, inference.py:126-128)
        self._queue.append((self._rows, query_tokens, kwargs))
        self._rows += rows

    def _pack(self):
        """Fill batch / mask from the queue: one scatter into the flattened arrays."""
        lengths = np.fromiter((len(q) for _, q, _ in self._queue), dtype=np.int64, count=len(self._queue))
        starts = np.fromiter((r for r, _, _ in self._queue), dtype=np.int64, count=len(self._queue)) * self.window_size
        ends = np.cumsum(lengths)
        # flat slot of every queued token: its query's first slot + its position in the query
        slots = np.repeat(starts - (ends - lengths), lengths) + np.arange(int(ends[-1]))
        self.batch.fill(0)
        self.mask.fill(0)
        self.batch.reshape(-1)[slots] = np.concatenate([np.asarray(q) for _, q, _ in self._queue])
        self.mask.reshape(-1)[slots] = 1
        return starts, lengths

    def process(self):
        if not self._queue:
            return
        logging.debug('Processing batch (batch size=%d, current batch=%d).', self.batch_size, self._rows)
        starts, lengths = self._pack()
        out = self.predict_fn(self.batch, self.mask)            # (B, n, V_e)
        per_token = out.reshape((-1, out.shape[-1]))
        for (_, payload, kwargs), s0, T in zip(self._queue, starts, lengths):
            self.callback(payload, per_token[s0:s0 + T], **kwargs)
        self._queue, self._rows = [], 0


class EmbeddingMapper(object):
    """One query at a time: mean word vector -> predict_fn -> callback."""

    def __init__(self, predict_fn, word_representations, result_callback):
        assert result_callback is None or callable(result_callback)
        self.predict_fn, self.callback = predict_fn, result_callback
        self.word_representations = word_representations

    def submit(self, query_tokens, **kwargs):
        # (sert/inference.py:161-167: the query is the mean of its words' rows)
        self.callback(query_tokens, self.predict_fn(self.word_representations[query_tokens, :].mean(axis=0)), **kwargs)

    def process(self):
        """Nothing is ever queued: submit() answers immediately."""


class BatchedEmbeddingMapper(object):
    """All queries at once (additive): ``submit`` queues, ``process`` projects
    every queued query with a single predict_fn call on the (Q, d_w) matrix of
    mean word vectors and hands the (Q, d_e) projections to
    ``callback.process_batch``."""

    def __init__(self, predict_fn, word_representations, result_callback):
        assert hasattr(result_callback, 'process_batch')
        self.predict_fn = predict_fn
        self.word_representations = word_representations
        self.callback = result_callback
        self.pending = []

    def submit(self, query_tokens, **kwargs):
        assert len(query_tokens) > 0
        self.pending.append((list(query_tokens), kwargs))

    def process(self):
        if not self.pending:
            return
        d = self.word_representations.shape[1]
        avg = np.empty((len(self.pending), d), dtype=np.float32)
        for i, (tokens, _) in enumerate(self.pending):
            avg[i] = self.word_representations[tokens, :].mean(axis=0)
        projections = self.predict_fn(avg)
        self.callback.process_batch(
            [p for p, _ in self.pending], projections,
            [kw for _, kw in self.pending])
        self.pending = []


def _product_skipping_zeros(distribution, axis):
    # multiplied in log space with log(0) counted as 0: a zero entry is skipped, it does not annihilate the product
    return np.exp(np.ma.log(distribution).filled(0).sum(axis=axis))


_AGGREGATORS = {
    'sum': lambda dist, axis: np.mean(dist, axis=axis),              # (sic: the reference's 'sum' is the mean)
    'product': _product_skipping_zeros,
    'last': lambda dist, axis: np.take(dist, indices=dist.shape[axis] - 1, axis=axis),
    'max': lambda dist, axis: np.max(dist, axis=axis),
    'identity': lambda dist, axis: dist,
}


def aggregate_distribution(distribution, mode, axis):
    """Per-token distributions -> one (sert/inference.py:170-183); unknown modes raise NotImplementedError."""
    try:
        fn = _AGGREGATORS[mode]
    except KeyError:
        raise NotImplementedError()
    return fn(distribution, axis)
