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
    assert result_callback is not None

    # id width of the packed batches (inference.py:10)
    instance_dtype = np.min_scalar_type(vocabulary_size - 1)
    logging.info('Instance elements will be stored using %s.', instance_dtype)

    if result_callback.should_average_input():
        if batched and hasattr(result_callback, 'process_batch'):
            return BatchedEmbeddingMapper(
                predict_fn, word_representations, result_callback)
        return EmbeddingMapper(
            predict_fn, word_representations, result_callback)

    return WordBatcher(
        predict_fn, batch_size, window_size, instance_dtype, result_callback)


class WordBatcher(object):
    """Fixed-shape batches for the loglinear predict_fn, which only accepts
    exactly (batch_size, window_size) inputs (models.py:830-856)."""

    OVERFLOW, TRUNCATE = range(5, 7)

    def __init__(self, predict_fn,
                 batch_size, window_size, instance_dtype,
                 result_callback=None,
                 overflow_mode=OVERFLOW):
        assert overflow_mode in (WordBatcher.OVERFLOW, WordBatcher.TRUNCATE)
        if result_callback is not None:
            assert hasattr(result_callback, '__call__')

        self.predict_fn = predict_fn
        self.batch_size = batch_size
        self.window_size = window_size
        self.overflow_mode = overflow_mode
        self.callback = result_callback

        self.batch = np.zeros((batch_size, window_size), dtype=instance_dtype)
        self.mask = np.zeros((batch_size, window_size), dtype=np.int8)

        self._empty_batch()

    def _empty_batch(self):
        self.batch.fill(0)     # padding rows/slots carry token id 0
        self.mask.fill(0)
        self.num_used_instances = 0
        self.requests = []

    def _rows_needed(self, num_tokens):
        return -(-num_tokens // self.window_size)

    def submit(self, query_tokens, **kwargs):
        assert len(query_tokens) > 0

        if self.overflow_mode == WordBatcher.TRUNCATE and \
                len(query_tokens) > self.window_size:
            logging.error('Truncated query "%s" as it exceeded '
                          'the window size.', query_tokens)
            query_tokens = query_tokens[:self.window_size]

        num_instances = self._rows_needed(len(query_tokens))

        if num_instances > self.batch_size:
            # a query longer than batch_size * window_size tokens cannot be
            # scored (inference.py:124-125)
            raise RuntimeError()
        if num_instances > self.batch_size - self.num_used_instances:
            self.process()

        self.requests.append((num_instances, query_tokens, kwargs))

        # long queries spill row-major over several rows (inference.py:132-143)
        first = self.num_used_instances
        flat = np.asarray(query_tokens)
        for r in range(num_instances):
            piece = flat[r * self.window_size:(r + 1) * self.window_size]
            self.batch[first + r, :len(piece)] = piece
            self.mask[first + r, :len(piece)] = 1
        self.num_used_instances += num_instances

    def process(self):
        if len(self.requests) == 0:
            return
        logging.debug('Processing batch (batch size=%d, current batch=%d).',
                      self.batch_size, self.num_used_instances)
        results = self.predict_fn(self.batch, self.mask)   # (B, n, V_e)

        row = 0
        for num_instances, payload, kwargs in self.requests:
            # the request's rows, flattened to one distribution per token and
            # cut back to the real token count (inference.py:89-94)
            result = results[row:row + num_instances]
            result = result.reshape((-1, result.shape[-1]))[:len(payload)]
            assert result.ndim == 2 and result.shape[0] == len(payload)

            self.callback(payload, result, **kwargs)
            row += num_instances

        self._empty_batch()


class EmbeddingMapper(object):
    """One query at a time: mean word vector -> predict_fn -> callback."""

    def __init__(self, predict_fn, word_representations, result_callback):
        if result_callback is not None:
            assert hasattr(result_callback, '__call__')

        self.predict_fn = predict_fn
        self.word_representations = word_representations
        self.callback = result_callback

    def process(self):
        return None   # (nothing is queued: submit() answers immediately)

    def submit(self, query_tokens, **kwargs):
        pooled = self.word_representations[query_tokens, :].mean(axis=0)
        projection = self.predict_fn(pooled)
        self.callback(query_tokens, projection, **kwargs)


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


def aggregate_distribution(distribution, mode, axis):
    """inference.py:170-183.  'product' multiplies in log space with log(0)
    treated as 0, i.e. zero entries are skipped, not annihilating."""
    if mode == 'sum':
        return np.mean(distribution, axis=axis)
    if mode == 'product':
        logs = np.ma.log(distribution).filled(0)
        return np.exp(np.sum(logs, axis=axis))
    if mode == 'last':
        return np.take(distribution, axis=axis,
                       indices=distribution.shape[axis] - 1)
    if mode == 'max':
        return np.max(distribution, axis=axis)
    if mode == 'identity':
        return distribution
    raise NotImplementedError()
