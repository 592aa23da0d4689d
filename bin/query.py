#!/usr/bin/env python
"""Query CLI -- drop-in for the reference's bin/query.py (same flags, same
files in and out): load a model dumped by bin/train.py, rank entities for every
topic, write TREC run files.

    python bin/query.py --meta meta --model model_15.bin --topics topics \
        --top 100 --run_out run

Writes ``<run_out>_debug``, ``<run_out>_ep`` (topics per entity) and
``<run_out>_ef`` (entities per topic) (query.py:94, :149-156).

Additive flags: --no_batch (score one query at a time like the reference
instead of all queries in one device call), --device.
"""
import argparse
import collections
import io
import logging
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sert_amd import inference, models, scoring  # noqa: E402
from sert_amd.utils import argparse_utils, logging_utils, trec_utils  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--loglevel', type=str, default='INFO')

    parser.add_argument('--meta', type=argparse_utils.existing_file_path, required=True)
    parser.add_argument('--model', type=argparse_utils.existing_file_path, required=True)

    parser.add_argument('--topics', type=argparse_utils.existing_file_path, nargs='+')

    parser.add_argument('--top', type=argparse_utils.positive_int, default=None)

    parser.add_argument('--run_out', type=argparse_utils.nonexisting_file_path,
                        required=True)

    # additive
    parser.add_argument('--no_batch', action='store_true', default=False)
    parser.add_argument('--device', type=int, default=0)
    return parser


def load_model(path):
    """[args, predict_fn, R_w, (R_e)] (train.py:289-300)."""
    with open(path, 'rb') as f:
        model_args, predict_fn = (pickle.load(f) for _ in range(2))

        word_representations = pickle.load(f)

        try:
            entity_representations = pickle.load(f)
        except EOFError:
            entity_representations = None
        if isinstance(entity_representations, dict):   # trailing optimiser-state pickle
            entity_representations = None
    return model_args, predict_fn, word_representations, entity_representations


def main(argv=None):
    args = build_parser().parse_args(argv)

    try:
        logging_utils.configure_logging(args)
    except IOError:
        return -1

    model_args, predict_fn, word_representations, entity_representations = \
        load_model(args.model)

    with open(args.meta, 'rb') as f:
        (data_args,
         words, tokens,
         entity_indices_inv, entity_assocs) = (
            pickle.load(f) for _ in range(5))

    topic_f = [open(filename, 'r') for filename in args.topics]
    topics = trec_utils.parse_topics(topic_f)
    for f_ in topic_f:
        f_.close()

    model_name = os.path.basename(args.model)

    # Entity profiling / entity finding.
    topics_per_entity = collections.defaultdict(list)
    entities_per_topic = collections.defaultdict(list)

    def ranker_callback(topic_id, top_ranked_indices, top_ranked_values):
        for entity_internal_id, relevance in zip(top_ranked_indices, top_ranked_values):
            entity_id = entity_indices_inv[entity_internal_id]

            topics_per_entity[entity_id].append((relevance, topic_id))
            entities_per_topic[topic_id].append((relevance, entity_id))

    with open('{0}_debug'.format(args.run_out), 'w') as f_debug_out:
        if model_args.type == models.LanguageModel:
            result_callback = scoring.LogLinearCallback(
                args, model_args, tokens, f_debug_out, ranker_callback)
        elif model_args.type in (models.VectorSpaceLanguageModel,
                                 models.VectorSpaceSoftmaxLanguageModel):
            result_callback = scoring.VectorSpaceCallback(
                entity_representations,
                args, model_args, tokens, f_debug_out, ranker_callback,
                device=args.device)
        else:
            raise RuntimeError('Unknown model type %s.' % model_args.type)

        batcher = inference.create(
            predict_fn, word_representations,
            model_args.batch_size, data_args.window_size, len(words),
            result_callback, batched=not args.no_batch)

        logging.info('Batching queries using %s.', batcher)

        for q_id, (topic_id, terms) in enumerate(topics.items()):
            query_terms = trec_utils.parse_query(terms)

            logging.debug('Query (%d/%d) %s: %s (%s)',
                          q_id + 1, len(topics), topic_id, query_terms, terms)

            query_tokens = []
            for term in query_terms:
                if term not in words:
                    logging.debug('Term "%s" is OOV.', term)
                    continue

                query_tokens.append(words[term].id)

            if not query_tokens:
                logging.warning('Skipping query with terms "%s".', terms)
                continue

            batcher.submit(query_tokens, topic_id=topic_id)

        batcher.process()

    with io.open('{0}_ep'.format(args.run_out), 'w', encoding='utf8') as out_ep_run:
        trec_utils.write_run(model_name, topics_per_entity, out_ep_run)

    with io.open('{0}_ef'.format(args.run_out), 'w', encoding='utf8') as out_ef_run:
        trec_utils.write_run(model_name, entities_per_topic, out_ef_run)

    logging.info('Saved run to %s.', args.run_out)


if __name__ == "__main__":
    sys.exit(main())
