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

Additive flags: --ignore_weights (the reference reads args.ignore_weights but
never defines it, train.py:81), --seed, --device, --save_optimizer_state.
Data-parallel: launch with ``python -m torch.distributed.run --nproc-per-node N``.
"""
import argparse
import logging
import os
import pickle
import sys

import numpy as np
import scipy
import scipy.sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sert_amd import distributed, models  # noqa: E402
from sert_amd.utils import argparse_utils, embedding_utils, logging_utils  # noqa: E402

MODELS = {
    'loglinear': models.LanguageModel,
    'vectorspace': models.VectorSpaceLanguageModel,
    # additive (not in the reference): vectorspace encoder + full softmax over entities
    'vectorspace_softmax': models.VectorSpaceSoftmaxLanguageModel,
}


def glorot_uniform(shape):
    """lasagne.init.GlorotUniform().sample (train.py:128-129, :170-171): U(-a, a),
    a = sqrt(6 / (fan_in + fan_out)), drawn from the global np.random."""
    a = np.sqrt(6.0 / (shape[0] + shape[1]))
    return np.random.uniform(low=-a, high=a, size=shape).astype(np.float32)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--loglevel', type=str, default='INFO')

    parser.add_argument('--data', type=argparse_utils.existing_file_path, required=True)
    parser.add_argument('--meta', type=argparse_utils.existing_file_path, required=True)

    parser.add_argument('--type', choices=sorted(MODELS), required=True)

    parser.add_argument('--iterations', type=argparse_utils.positive_int, default=1)
    parser.add_argument('--batch_size', type=argparse_utils.positive_int, default=1024)

    parser.add_argument('--word_representation_size',
                        type=argparse_utils.positive_int, default=300)
    parser.add_argument('--representation_initializer',
                        type=argparse_utils.existing_file_path, default=None)

    # Specific to VectorSpaceLanguageModel.
    parser.add_argument('--entity_representation_size',
                        type=argparse_utils.positive_int, default=None)
    parser.add_argument('--num_negative_samples',
                        type=argparse_utils.positive_int, default=None)
    parser.add_argument('--one_hot_classes', action='store_true', default=False)

    parser.add_argument('--regularization_lambda', type=argparse_utils.ratio, default=0.01)

    parser.add_argument('--model_output', type=str, required=True)

    # additive
    parser.add_argument('--ignore_weights', action='store_true', default=False)
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--device', type=int, default=None)
    parser.add_argument('--save_optimizer_state', action='store_true', default=False)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)

    if args.entity_representation_size is None:
        args.entity_representation_size = args.word_representation_size

    args.type = MODELS[args.type]

    try:
        logging_utils.configure_logging(args)
    except IOError:
        return -1

    logging_utils.log_module_info(np, scipy)

    ctx = distributed.init_from_env()
    if args.seed is not None:
        np.random.seed(args.seed)
    elif ctx.world_size > 1:
        np.random.seed(distributed.broadcast_object(int(np.random.randint(1 << 30))))

    logging.info('Loading data from %s.', args.data)
    data_sets = np.load(args.data, allow_pickle=True)

    if 'w_train' in data_sets and not args.ignore_weights:
        w_train = data_sets['w_train']
    else:
        logging.warning('No weights found in data set; '
                        'assuming uniform instance weighting.')

        w_train = np.ones(data_sets['x_train'].shape[0], dtype=np.float32)

    training_set = (data_sets['x_train'], data_sets['y_train'][()], w_train)
    validation_set = (data_sets['x_validate'], data_sets['y_validate'][()])

    logging.info('Training instances: %s (%s) %s (%s) %s (%s)',
                 training_set[0].shape, training_set[0].dtype,
                 training_set[1].shape, training_set[1].dtype,
                 training_set[2].shape, training_set[2].dtype)
    logging.info('Validation instances: %s (%s) %s (%s)',
                 validation_set[0].shape, validation_set[0].dtype,
                 validation_set[1].shape, validation_set[1].dtype)

    num_entities = training_set[1].shape[1]
    assert num_entities > 1

    if args.one_hot_classes:
        logging.info('Transforming y-values to one-hot values.')

        if not scipy.sparse.issparse(training_set[1]) or \
           not scipy.sparse.issparse(validation_set[1]):
            raise RuntimeError(
                'Argument --one_hot_classes expects sparse truth values.')

        y_train, (x_train, w_train) = sparse_to_one_hot_multiple(
            training_set[1], training_set[0], training_set[2])
        training_set = (x_train, y_train, w_train)

        y_validate, (x_validate,) = sparse_to_one_hot_multiple(
            validation_set[1], validation_set[0])
        validation_set = (x_validate, y_validate)

    logging.info('Loading meta-data from %s.', args.meta)
    with open(args.meta, 'rb') as f:
        # the rest of the stream (entity maps) is only needed at query time
        data_args, words, tokens = (pickle.load(f) for _ in range(3))

        vocabulary_size = len(words)

    representations = glorot_uniform(
        (vocabulary_size, args.word_representation_size))

    if args.representation_initializer:
        # later duplicates of a word in the initialiser file win (dict semantics)
        representation_lookup = dict(
            embedding_utils.load_binary_representations(
                args.representation_initializer, tokens))

        representation_init_count = 0

        for word, meta in words.items():
            if word.lower() in representation_lookup:
                representations[meta.id] = representation_lookup[word.lower()]

                representation_init_count += 1

        logging.info('Initialized representations from '
                     'pre-learned collection for %d words (%.2f%%).',
                     representation_init_count,
                     (representation_init_count / float(len(words))) * 100.0)

    del words
    del tokens

    model_options = {
        'batch_size': args.batch_size,
        'window_size': data_args.window_size,
        'representations_init': distributed.broadcast_array(representations),
        'regularization_lambda': args.regularization_lambda,
        'training_set': training_set,
        'validation_set': validation_set,
    }

    if args.type == models.LanguageModel:
        model_options.update(output_layer_size=num_entities)
    elif args.type in (models.VectorSpaceLanguageModel,
                       models.VectorSpaceSoftmaxLanguageModel):
        entity_representations = glorot_uniform(
            (num_entities, args.entity_representation_size))

        model_options.update(
            entity_representations_init=distributed.broadcast_array(entity_representations),
            num_negative_samples=args.num_negative_samples)

    if args.device is not None:
        args.type.device = args.device

    model = args.type(**model_options)

    train(model, args.iterations, args.model_output,
          abort_threshold=1e-5,
          early_stopping=False,
          additional_args=[args],
          save_optimizer_state=args.save_optimizer_state)

    distributed.shutdown()


def sparse_to_one_hot_multiple(y, *matrices):
    """One (instance, entity) pair per non-zero of the sparse truth matrix y
    (train.py:186-245): returns the int32 entity ids in row-major order and, for
    every extra matrix, its rows repeated once per non-zero of the same row.
    Every row must own at least one non-zero."""
    assert scipy.sparse.issparse(y), 'Matrix y should be sparse.'

    num_instances, num_classes = y.shape

    assert num_classes < (1 << 31), \
        'Number of classes should be encodable in 32-bit signed integer.'

    for matrix in matrices:
        assert isinstance(matrix, np.ndarray), \
            'Matrix {0} should be dense.'.format(repr(matrix))

        assert matrix.shape[0] == num_instances

    coo = y.tocoo()
    rows, cols = coo.row, coo.col

    # rows must appear as 0, 0.., 1, 1.., 2, ... without gaps (train.py:226-236)
    if rows.size:
        steps = np.diff(np.concatenate(([-1], rows)))
        if np.any((steps != 0) & (steps != 1)):
            raise RuntimeError(
                'Every truth value should have at least '
                'one non-zero index.')

    new_y = np.array(cols, dtype=np.int32)
    new_matrices = [np.array(matrix[rows], dtype=matrix.dtype) for matrix in matrices]

    return new_y, new_matrices


#
# Training driver.
#

def error_delta(error):
    if len(error) <= 1:
        return 0.0, 0.0

    absolute = error[-1] - error[-2]
    relative = absolute / float(error[-2])

    return absolute, relative


def train(model, num_epochs, output_path,
          abort_threshold=1e-5, early_stopping=False,
          additional_args=[], save_optimizer_state=False):
    """Epoch driver (train.py:262-348): errors before any training, dump of
    epoch 0, then per epoch train -> errors -> dump; stop when the training
    error moves by less than `abort_threshold`."""
    assert isinstance(model, models.ModelInterface)
    assert isinstance(abort_threshold, float)

    is_writer = distributed.get_context().rank == 0

    error_means = {'training': [], 'validation': []}
    error_stddevs = {'training': [], 'validation': []}

    def compute_errors():
        train_error_mean, train_error_std = model.train_error()
        validation_error_mean, validation_error_std = model.validation_error()

        error_means['training'].append(train_error_mean)
        error_means['validation'].append(validation_error_mean)

        error_stddevs['training'].append(train_error_std)
        error_stddevs['validation'].append(validation_error_std)

    def dump_model(epoch):
        state = list(model.get_state())    # collective-free, but every rank reads
        if not is_writer:
            return

        output_model_filename = '{0}_{1}.bin'.format(output_path, epoch)

        with open(output_model_filename, 'wb') as f:
            for obj in list(additional_args) + state:
                pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
            if save_optimizer_state and hasattr(model, 'get_optimizer_state'):
                # trailing pickle: old readers stop after 4 loads (query.py:54-62)
                pickle.dump({'optimizer_state': model.get_optimizer_state()}, f,
                            protocol=pickle.HIGHEST_PROTOCOL)

        model_size = os.path.getsize(output_model_filename)

        logging.info('Saved model "%s" (%d megabyte).',
                     output_model_filename, (model_size / 1024 / 1024))

    compute_errors()

    dump_model(0)

    for epoch in range(1, num_epochs + 1):
        logging.info('Epoch %d.', epoch)

        num_batches, mean_cost = model.train()

        logging.info('Epoch %d: processed %d batches; average error=%f.',
                     epoch, num_batches, mean_cost)

        logging.info('Epoch %d: measuring training/validation error.', epoch)

        compute_errors()

        logging.info('Training errors: %s; delta=%s',
                     list(zip(error_means['training'], error_stddevs['training'])),
                     error_delta(error_means['training']))
        logging.info('Validation errors: %s; delta=%s',
                     list(zip(error_means['validation'], error_stddevs['validation'])),
                     error_delta(error_means['validation']))

        dump_model(epoch=epoch)

        assert np.all(np.isfinite(error_means['training'][-1]))

        if early_stopping:
            assert np.all(np.isfinite(error_means['validation'][-1]))

            if error_means['validation'][-1] > error_means['validation'][-2]:
                logging.info('Validation error stopped decreasing; aborting.')

                return

        if len(error_means['training']) > 1 and \
                abs(error_means['training'][-1] -
                    error_means['training'][-2]) < abort_threshold:
            logging.error('No learning was performed during '
                          'the last iteration; aborting.')

            return


if __name__ == "__main__":
    sys.exit(main())
