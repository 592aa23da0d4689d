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
from sert_amd.utils import argparse_utils as au  # noqa: E402
from sert_amd.utils import logging_utils, trec_utils  # noqa: E402

VECTORSPACE_TYPES = (models.VectorSpaceLanguageModel, models.VectorSpaceSoftmaxLanguageModel)


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    parser.add_argument('--loglevel', type=str, default='INFO')
    parser.add_argument('--meta', type=au.existing_file_path, required=True)
    parser.add_argument('--model', type=au.existing_file_path, required=True)
    parser.add_argument('--topics', type=au.existing_file_path, nargs='+')
    parser.add_argument('--top', type=au.positive_int, default=None)
    parser.add_argument('--run_out', type=au.nonexisting_file_path, required=True)
    parser.add_argument('--no_batch', action='store_true', default=False)
    parser.add_argument('--device', type=int, default=0)
    return parser


def load_model(path):
    """Pickle stream of bin/train.py: namespace, predict_fn, R_w[, R_e][, extras]."""
    objects = []
    with open(path, 'rb') as f:
        while len(objects) < 5:
            try:
                objects.append(pickle.load(f))
            except EOFError:
                break
    model_args, predict_fn, word_representations = objects[:3]
    entity_representations = None
    if len(objects) > 3 and not isinstance(objects[3], dict):   # dict = optimiser-state trailer
        entity_representations = objects[3]
    return model_args, predict_fn, word_representations, entity_representations


def load_meta(path):
    """Five pickles written by bin/prepare.py:373-376."""
    with open(path, 'rb') as f:
        return tuple(pickle.load(f) for _ in range(5))


def topic_tokens(text, words):
    """In-vocabulary token ids of a topic; OOV terms are dropped (query.py:129-137)."""
    ids = []
    for term in trec_utils.parse_query(text):
        entry = words.get(term)
        if entry is None:
            logging.debug('Term "%s" is OOV.', term)
        else:
            ids.append(entry.id)
    return ids


def main(argv=None):
    args = build_parser().parse_args(argv)
    try:
        logging_utils.configure_logging(args)
    except IOError:
        return -1

    model_args, predict_fn, word_representations, entity_representations = load_model(args.model)
    data_args, words, tokens, entity_indices_inv, _entity_assocs = load_meta(args.meta)

    handles = [open(name, 'r') for name in args.topics]
    try:
        topics = trec_utils.parse_topics(handles)
    finally:
        for handle in handles:
            handle.close()

    by_entity = collections.defaultdict(list)   # entity profiling: topics per entity
    by_topic = collections.defaultdict(list)    # entity finding: entities per topic

    def collect(topic_id, entity_indices, scores):
        for internal_id, score in zip(entity_indices, scores):
            entity_id = entity_indices_inv[internal_id]
            by_entity[entity_id].append((score, topic_id))
            by_topic[topic_id].append((score, entity_id))

    with open(args.run_out + '_debug', 'w') as debug_out:
        if model_args.type is models.LanguageModel:
            callback = scoring.LogLinearCallback(args, model_args, tokens, debug_out, collect)
        elif model_args.type in VECTORSPACE_TYPES:
            callback = scoring.VectorSpaceCallback(entity_representations, args, model_args, tokens,
                                                   debug_out, collect, device=args.device)
        else:
            raise RuntimeError('Unknown model type %s.' % model_args.type)

        batcher = inference.create(predict_fn, word_representations, model_args.batch_size,
                                   data_args.window_size, len(words), callback,
                                   batched=not args.no_batch)
        logging.info('Batching queries using %s.', batcher)

        for position, (topic_id, text) in enumerate(topics.items(), 1):
            ids = topic_tokens(text, words)
            logging.debug('Query (%d/%d) %s: %s', position, len(topics), topic_id, text)
            if ids:
                batcher.submit(ids, topic_id=topic_id)
            else:
                logging.warning('Skipping query with terms "%s".', text)
        batcher.process()

    run_name = os.path.basename(args.model)
    for suffix, ranking in (('_ep', by_entity), ('_ef', by_topic)):
        with io.open(args.run_out + suffix, 'w', encoding='utf8') as out:
            trec_utils.write_run(run_name, ranking, out)
    logging.info('Saved run to %s.', args.run_out)


if __name__ == "__main__":
    sys.exit(main())
