"""Ranker callbacks of the query path (the reference keeps them in
bin/query.py:165-382): they receive what ``sert.inference`` produced for one
query and turn it into a ranking of entities.

  LogLinearCallback    product of the per-token entity distributions ->
                       renormalise -> rank ALL entities (query.py:199-236)
  VectorSpaceCallback  cosine scoring of the query projection against the
                       L2-normalised entity table, top-k (query.py:239-370);
                       here the normalisation, the (Q,d)x(d,V_e) scoring GEMM and
                       the top-k selection run on the MI355X (sert_scorer_* in
                       include/sert_hip.h); ``process_batch`` scores all queries
                       of a run in one call.
"""
import logging

import numpy as np

from sert_amd import _capi, inference, math_utils


class Callback(object):
    """query.py:165-196."""

    def __init__(self, args, model_args, tokens,
                 f_debug_out,
                 rank_callback):
        self.args = args
        self.model_args = model_args

        self.tokens = tokens

        self.f_debug_out = f_debug_out

        self.rank_callback = rank_callback

        self.topic_projections = {}

    def __call__(self, payload, result, topic_id):
        assert topic_id not in self.topic_projections
        self.topic_projections[topic_id] = result.ravel()

        logging.debug('Result of shape %s for topic "%s".',
                      result.shape, topic_id)

        self.process(payload, result, topic_id)

    def process(self, payload, distribution, topic_id):
        raise NotImplementedError()

    def should_average_input(self):
        raise NotImplementedError()


def compute_normalised_entropy(distribution, base=2):
    """query.py:373-382."""
    assert distribution.ndim == 2

    assert np.allclose(distribution.sum(axis=1), 1.0)

    return [math_utils.entropy(distribution[i, :], base=base, normalize=True)
            for i in range(distribution.shape[0])]


class LogLinearCallback(Callback):

    def process(self, payload, distribution, topic_id):
        terms = [self.tokens[token_id] for token_id in payload]
        term_entropies = compute_normalised_entropy(distribution, base=2)

        # P(e | q) ~ prod_t P(e | w_t)  (query.py:209-214)
        distribution = inference.aggregate_distribution(
            distribution, mode='product', axis=0)

        assert distribution.ndim == 1

        distribution /= distribution.sum()

        if not np.isclose(distribution.sum(), 1.0):
            logging.error('Encountered non-normalized '
                          'distribution for topic "%s" '
                          '(mass=%.10f).',
                          topic_id, distribution.sum())

        if self.f_debug_out is not None:
            self.f_debug_out.write('Topic {0} {1}: {2}\n'.format(
                topic_id,
                math_utils.entropy(distribution, base=2, normalize=True),
                list(zip(terms, term_entropies))))

        # ascending argsort, reversed: every entity is ranked (query.py:228-231)
        top_ranked_indices = np.argsort(distribution)[::-1]
        top_ranked_values = distribution[top_ranked_indices]

        self.rank_callback(topic_id, top_ranked_indices, top_ranked_values)

    def should_average_input(self):
        return False


class VectorSpaceCallback(Callback):

    def __init__(self, entity_representations, *args, **kwargs):
        self.device = kwargs.pop('device', 0)
        super(VectorSpaceCallback, self).__init__(*args, **kwargs)

        logging.info(
            'Initializing k-NN for entity representations of shape %s.',
            entity_representations.shape)

        num_entities = entity_representations.shape[0]
        n_neighbors = self.args.top

        if n_neighbors is None:
            logging.warning(
                'Parameter k not set; defaulting to all entities (k=%d).',
                num_entities)
        elif n_neighbors > num_entities:
            logging.warning(
                'Parameter k exceeds number of entities; '
                'defaulting to all entities (k=%d).',
                num_entities)

            n_neighbors = None

        self.n_neighbors = n_neighbors

        # cosine == euclidean on L2-normalised vectors (query.py:262-274); the
        # device copy is normalised, the caller's array is left untouched
        self.normalize_representations = True
        self.scorer = _capi.Scorer(entity_representations, device=self.device)

        logging.info('Using cosine scoring on the MI355X (%s).',
                     _capi.device_info(self.device))

    def _check_projection(self, term_projections):
        _, entity_representation_size = term_projections.shape
        assert(entity_representation_size ==
               self.model_args.entity_representation_size)

    def process(self, payload, result, topic_id):
        term_projections = inference.aggregate_distribution(
            result, mode='identity', axis=0)

        if term_projections.ndim == 1:
            term_projections = term_projections.reshape(1, -1)

        self._check_projection(term_projections)

        # one centroid per query (query.py:346)
        assert term_projections.shape[0] == 1

        indices, values = self.scorer.rank(term_projections, self.n_neighbors)

        self.rank_callback(topic_id, indices[0].astype(np.int64), values[0])

    def process_batch(self, payloads, projections, kwargs_list):
        """All queries of a run in one scoring call (additive)."""
        projections = np.asarray(projections, dtype=np.float32)
        projections = projections.reshape(len(payloads), -1)
        self._check_projection(projections)
        for kw, p in zip(kwargs_list, projections):
            topic_id = kw['topic_id']
            assert topic_id not in self.topic_projections
            self.topic_projections[topic_id] = p.ravel()
        indices, values = self.scorer.rank(projections, self.n_neighbors)
        for qi, kw in enumerate(kwargs_list):
            self.rank_callback(kw['topic_id'], indices[qi].astype(np.int64),
                               values[qi])

    def should_average_input(self):
        return True
