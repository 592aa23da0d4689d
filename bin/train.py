#!/usr/bin/env python
"""Training CLI -- drop-in for the reference's bin/train.py (same flags, same
files in and out), executed on MI355X through sert_amd.models.

    python bin/train.py --data data.npz --meta meta --type vectorspace \
        --batch_size 4096 --word_representation_size 300 \
        --entity_representation_size 128 --num_negative_samples 10 \
        --one_hot_classes --iterations 15 --model_output model

Inputs (bin/prepare.py:372-416): ``data.npz`` with x_train (N,n) uint,
y_train (0-d object array holding a csr_matrix (N,V_e) f32), optional w_train,
x_validate, y_validate; ``meta`` = pickle stream (args, words, tokens, ...).
Output: ``<model_output>_<epoch>.bin`` = pickle stream [args, predict_fn, R_w,
(R_e)] (bin/train.py:289-300) readable by bin/query.py.

Additive: --type vectorspace_softmax, --ignore_weights (the reference reads
args.ignore_weights but never defines it, train.py:81), --seed, --device,
--save_optimizer_state, --resume <model_k.bin> (continue behind epoch k of a run that was
dumped with --save_optimizer_state: same results as the uninterrupted run), --gpus N
(data parallel over N GPUs of this node, one process each, started by this script;
--batch_size is the GLOBAL batch).  Any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE
works as well.
"""
import argparse
import logging
import os
import pickle
import sys

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sert_amd import distributed, models, training  # noqa: E402
from sert_amd.training import error_delta, sparse_to_one_hot_multiple, train  # noqa: E402,F401
from sert_amd.utils import argparse_utils as au  # noqa: E402
from sert_amd.utils import embedding_utils, logging_utils  # noqa: E402

MODELS = {
    'loglinear': models.LanguageModel,
    'vectorspace': models.VectorSpaceLanguageModel,
    # additive (not in the reference): vectorspace encoder + full softmax over entities
    'vectorspace_softmax': models.VectorSpaceSoftmaxLanguageModel,
}

# (flag, keyword arguments) -- the reference's surface first (train.py:28-61)
FLAGS = [
    ('--loglevel', dict(type=str, default='INFO')),
    ('--data', dict(type=au.existing_file_path, required=True)),
    ('--meta', dict(type=au.existing_file_path, required=True)),
    ('--type', dict(choices=sorted(MODELS), required=True)),
    ('--iterations', dict(type=au.positive_int, default=1)),
    ('--batch_size', dict(type=au.positive_int, default=1024)),
    ('--word_representation_size', dict(type=au.positive_int, default=300)),
    ('--representation_initializer', dict(type=au.existing_file_path, default=None)),
    ('--entity_representation_size', dict(type=au.positive_int, default=None)),
    ('--num_negative_samples', dict(type=au.positive_int, default=None)),
    ('--one_hot_classes', dict(action='store_true', default=False)),
    ('--regularization_lambda', dict(type=au.ratio, default=0.01)),
    ('--model_output', dict(type=str, required=True)),
    # additive
    ('--ignore_weights', dict(action='store_true', default=False)),
    ('--seed', dict(type=int, default=None)),
    ('--device', dict(type=int, default=None)),
    ('--save_optimizer_state', dict(action='store_true', default=False)),
    ('--resume', dict(type=au.existing_file_path, default=None)),
    ('--gpus', dict(type=au.positive_int, default=1)),
]


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for flag, kwargs in FLAGS:
        parser.add_argument(flag, **kwargs)
    return parser


def glorot_uniform(shape):
    """U(-a, a), a = sqrt(6 / (rows + cols)), from the global np.random -- what
    lasagne.init.GlorotUniform().sample gives at train.py:128-129 and :170-171."""
    limit = np.sqrt(6.0 / sum(shape))
    return np.random.uniform(-limit, limit, size=shape).astype(np.float32)


def initial_word_representations(args, words, tokens):
    table = glorot_uniform((len(words), args.word_representation_size))
    if not args.representation_initializer:
        return table
    # pre-trained vectors override the random rows; for duplicated words in the
    # initialiser file the last one wins (train.py:131-151)
    pretrained = dict(embedding_utils.load_binary_representations(
        args.representation_initializer, tokens))
    hits = 0
    for word, entry in words.items():
        vector = pretrained.get(word.lower())
        if vector is not None:
            table[entry.id] = vector
            hits += 1
    logging.info('Initialized representations from pre-learned collection '
                 'for %d words (%.2f%%).', hits, 100.0 * hits / float(len(words)))
    return table


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.entity_representation_size is None:
        args.entity_representation_size = args.word_representation_size
    # the model class itself travels in the dumped namespace (train.py:68)
    args.type = MODELS[args.type]

    try:
        logging_utils.configure_logging(args)
    except IOError:
        return -1
    logging_utils.log_module_info(np, scipy)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # one process per GPU, started here; this process only waits for them
        command = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
        return distributed.launch(command, args.gpus)

    ctx = distributed.init_from_env()
    if args.seed is not None:
        np.random.seed(args.seed)
    elif ctx.world_size > 1:
        np.random.seed(distributed.broadcast_object(int(np.random.randint(1 << 30))))

    training_set, validation_set = training.load_data_sets(args.data, args.ignore_weights)
    num_entities = training_set[1].shape[1]
    assert num_entities > 1
    if args.one_hot_classes:
        training_set, validation_set = training.to_one_hot(training_set, validation_set)

    logging.info('Loading meta-data from %s.', args.meta)
    with open(args.meta, 'rb') as f:
        # args of prepare, word -> entry(.id), id -> word; the entity maps that
        # follow in the stream are only needed at query time
        data_args, words, tokens = (pickle.load(f) for _ in range(3))

    options = dict(
        batch_size=args.batch_size,
        window_size=data_args.window_size,
        representations_init=distributed.broadcast_array(
            initial_word_representations(args, words, tokens)),
        regularization_lambda=args.regularization_lambda,
        training_set=training_set,
        validation_set=validation_set)
    del words, tokens

    if args.type is models.LanguageModel:
        options['output_layer_size'] = num_entities
    else:
        options['entity_representations_init'] = distributed.broadcast_array(
            glorot_uniform((num_entities, args.entity_representation_size)))
        options['num_negative_samples'] = args.num_negative_samples

    if args.device is not None:
        args.type.device = args.device
    checkpoint = None
    if args.resume:
        checkpoint = training.read_checkpoint(args.resume)
        if checkpoint['trailer'] is not None:   # the sampler seed is fixed at construction
            args.type.sampler_seed = checkpoint['trailer']['sampler_state']['seed']
    model = args.type(**options)
    if checkpoint is not None:
        training.restore(model, checkpoint)

    train(model, args.iterations, args.model_output,
          abort_threshold=1e-5, early_stopping=False,
          additional_args=[args],
          save_optimizer_state=args.save_optimizer_state,
          resume_from=checkpoint)

    distributed.shutdown()


if __name__ == "__main__":
    sys.exit(main())
