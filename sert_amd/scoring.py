"""Ranker callbacks of the query path.  The reference keeps them inside
bin/query.py (lines 165-382); here they are importable.  A callback receives
what ``sert.inference`` produced for one query -- ``callback(payload, result,
topic_id=...)`` -- and reports a ranking through
``rank_callback(topic_id, entity_indices, scores)`` (best first).

  LogLinearCallback    result = (T, V_e) per-token entity distributions.
                       Score = product over the tokens (zeros skipped),
                       renormalised; EVERY entity is ranked (query.py:199-236).
  VectorSpaceCallback  result = (1, d_e) query projection.  Score of an entity =
                       (cosine + 1) / 2 against the L2-normalised entity table;
                       the best --top entities (or all) are ranked
                       (query.py:239-370).  Normalisation, the scoring GEMM and
                       the top-k selection run on the MI355X (sert_scorer_* of
                       include/sert_hip.h); ``process_batch`` scores all queries
                       of a run in ONE device call.
"""
import logging

import numpy as np

from sert_amd import _capi, inference, math_utils


def compute_normalised_entropy(distribution, base=2):
    """Normalised Shannon entropy of every row of a (T, V_e) matrix of
    distributions (each row must sum to one)."""
    assert distribution.ndim == 2
    assert np.allclose(distribution.sum(axis=1), 1.0)
    return [math_utils.entropy(row, base=base, normalize=True) for row in distribution]


class Callback(object):
    """Shared bookkeeping: remembers the raw result per topic and refuses to see
    a topic twice."""

    def __init__(self, args, model_args, tokens, f_debug_out, rank_callback):
        self.args, self.model_args = args, model_args
        self.tokens = tokens
        self.f_debug_out = f_debug_out
        self.rank_callback = rank_callback
        self.topic_projections = {}

    def _remember(self, topic_id, result):
        assert topic_id not in self.topic_projections
        self.topic_projections[topic_id] = np.ravel(result)

    def __call__(self, payload, result, topic_id):
        self._remember(topic_id, result)
        logging.debug('Result of shape %s for topic "%s".', result.shape, topic_id)
        self.process(payload, result, topic_id)

    def process(self, payload, distribution, topic_id):
        raise NotImplementedError()

    def should_average_input(self):
        raise NotImplementedError()


class LogLinearCallback(Callback):

    def should_average_input(self):
        return False

    def process(self, payload, distribution, topic_id):
        per_token_entropy = compute_normalised_entropy(distribution, base=2)

        joint = inference.aggregate_distribution(distribution, mode='product', axis=0)
        assert joint.ndim == 1
        joint /= joint.sum()
        mass = joint.sum()
        if not np.isclose(mass, 1.0):
            logging.error('Encountered non-normalized distribution for topic "%s" '
                          '(mass=%.10f).', topic_id, mass)

        if self.f_debug_out is not None:
            words = [self.tokens[token] for token in payload]
            self.f_debug_out.write('Topic {0} {1}: {2}\n'.format(
                topic_id, math_utils.entropy(joint, base=2, normalize=True),
                list(zip(words, per_token_entropy))))

        # ascending argsort read backwards = best first; all V_e entities are ranked
        order = np.argsort(joint)[::-1]
        self.rank_callback(topic_id, order, joint[order])


class VectorSpaceCallback(Callback):

    def __init__(self, entity_representations, *args, **kwargs):
        self.device = kwargs.pop('device', 0)
        super(VectorSpaceCallback, self).__init__(*args, **kwargs)

        num_entities = entity_representations.shape[0]
        logging.info('Initializing k-NN for entity representations of shape %s.',
                     entity_representations.shape)

        k = self.args.top
        if k is None:
            logging.warning('Parameter k not set; defaulting to all entities (k=%d).',
                            num_entities)
        elif k > num_entities:
            logging.warning('Parameter k exceeds number of entities; '
                            'defaulting to all entities (k=%d).', num_entities)
            k = None
        self.n_neighbors = k

        # cosine ranking == euclidean k-NN on unit vectors; the device copy of the
        # table is normalised, the caller's array is left as it was
        self.normalize_representations = True
        self.scorer = _capi.Scorer(entity_representations, device=self.device)
        logging.info('Cosine scoring on %s.', _capi.device_info(self.device))

    def should_average_input(self):
        return True

    def _as_queries(self, projections, count):
        q = np.asarray(projections, dtype=np.float32).reshape(count, -1)
        assert q.shape[1] == self.model_args.entity_representation_size
        return q

    def process(self, payload, result, topic_id):
        identity = inference.aggregate_distribution(result, mode='identity', axis=0)
        rows = 1 if identity.ndim == 1 else identity.shape[0]
        assert rows == 1, 'one centroid per query'
        idx, score = self.scorer.rank(self._as_queries(identity, 1), self.n_neighbors)
        self.rank_callback(topic_id, idx[0].astype(np.int64), score[0])

    def process_batch(self, payloads, projections, kwargs_list):
        """Additive: all queued queries at once (the query block is built in the scorer's
        page-locked buffer, so it is uploaded without a staging copy)."""
        queries = self.scorer.query_buffer(len(payloads))
        np.copyto(queries, self._as_queries(projections, len(payloads)))
        for kw, q in zip(kwargs_list, queries):
            # (a copy: `queries` is the scorer's reusable page-locked block, overwritten by the next call)
            self._remember(kw['topic_id'], q.copy())
        idx, score = self.scorer.rank(queries, self.n_neighbors)
        for row, kw in enumerate(kwargs_list):
            self.rank_callback(kw['topic_id'], idx[row].astype(np.int64), score[row])
