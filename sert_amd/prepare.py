"""Instance generation: TREC-text documents + entity associations -> ``data.npz``
+ ``meta`` in the layouts ``bin/train.py`` / ``bin/query.py`` consume
(written by the reference's bin/prepare.py:372-416).

CPU text preprocessing -- not on the GPU hot path ("the step
before the path").  Behaviour follows bin/prepare.py; the tokenisation /
windowing helpers the reference takes from the un-vendored ``cvangysel-common``
are restated from their call sites (prepare.py:496-512) and documented here:

  * a document's text is lower-cased, every non-alphanumeric character is a
    separator, purely numeric tokens become ``<num>``;
  * the vocabulary keeps words with >= min_count occurrences and
    >= min_word_size characters that are not in the ignore list, the
    ``max_size`` most frequent ones (ties: alphabetical), ids by rank;
    with padding enabled the padding token ``</s>`` is part of it;
  * a document becomes windows of ``window_size`` in-vocabulary token ids taken
    every ``stride`` tokens (default stride = window size; ``--overlapping`` =
    stride 1); the last, incomplete window is right-padded with ``</s>``
    (``--no_padding``: dropped);
  * every window is labelled with the uniform distribution over the entities
    associated with its document (prepare.py:433-436, :519-523);
  * instance weight = (longest document, in windows) / (windows of the
    instance's document) (prepare.py:394-397);
  * ``--resample``: every distinct label set is re-sampled (with replacement)
    to the average number of instances per label set (prepare.py:265-304).
"""
import collections
import logging
import pickle
import re

import numpy as np
import scipy.sparse as sparse

PADDING_TOKEN = '</s>'
NUMERIC_TOKEN = '<num>'

# A compact English stop list standing in for nltk's (nltk is not available
# offline); --remove_stopwords <file> supplies any other list.
BUILTIN_STOPWORDS = frozenset('''a about above after again against all am an and any are as at be because been
before being below between both but by can did do does doing down during each few for from further had has have
having he her here hers herself him himself his how i if in into is it its itself just me more most my myself no nor
not now of off on once only or other our ours ourselves out over own same she should so some such than that the
their theirs them themselves then there these they this those through to too under until up very was we were what
when where which while who whom why will with you your yours yourself yourselves'''.split())

MARKUP_TOKENS = ('<doc>', '</doc>', '<docno>', '<text>', '</text>')

Word = collections.namedtuple('Word', ['id', 'count'])

_DOC = re.compile(r'<DOC>(.*?)</DOC>', re.S | re.I)
_DOCNO = re.compile(r'<DOCNO>\s*(.*?)\s*</DOCNO>', re.S | re.I)
_TEXT = re.compile(r'<TEXT>(.*?)</TEXT>', re.S | re.I)
_TAG = re.compile(r'<[^>]+>')
_SEP = re.compile(r'[^0-9a-z]+')


def iter_trec_documents(paths, encoding='latin1'):
    """Yield (document id, text) from TREC-text files."""
    for path in paths:
        with open(path, 'r', encoding=encoding) as f:
            content = f.read()
        for body in _DOC.findall(content):
            docno = _DOCNO.search(body)
            if not docno:
                continue
            texts = _TEXT.findall(body)
            text = ' '.join(texts) if texts else _DOCNO.sub(' ', body)
            yield docno.group(1), _TAG.sub(' ', text)


def tokenize(text):
    """Lower-case alphanumeric tokens; numbers collapse to ``<num>``."""
    return [NUMERIC_TOKEN if tok.isdigit() else tok
            for tok in _SEP.split(text.lower()) if tok]


def extract_vocabulary(documents, min_count=2, max_size=65536, min_word_size=2,
                       ignore=(), with_padding=True):
    """-> (words: word -> Word(id, count), tokens: id -> word)."""
    ignore = set(ignore)
    counts = collections.Counter()
    for _, text in documents:
        counts.update(tok for tok in tokenize(text)
                      if tok not in ignore and len(tok) >= min_word_size)
    kept = [(w, c) for w, c in counts.items() if c >= min_count]
    kept.sort(key=lambda wc: (-wc[1], wc[0]))
    room = max_size - (1 if with_padding else 0)
    kept = kept[:max(0, room)]
    words, tokens = {}, {}
    if with_padding:
        words[PADDING_TOKEN] = Word(0, 0)
        tokens[0] = PADDING_TOKEN
    for w, c in kept:
        idx = len(words)
        words[w] = Word(idx, c)
        tokens[idx] = w
    return words, tokens


def read_associations(f, known_documents=None):
    """Lines ``entity_id document_id [1]`` -> (entities_per_document,
    documents_per_entity); associations to unknown documents are dropped."""
    entities_per_document = collections.defaultdict(list)
    documents_per_entity = collections.defaultdict(list)
    for line in f:
        parts = line.split()
        if len(parts) < 2:
            continue
        entity_id, document_id = parts[0], parts[1]
        if known_documents is not None and document_id not in known_documents:
            continue
        if entity_id not in entities_per_document[document_id]:
            entities_per_document[document_id].append(entity_id)
            documents_per_entity[entity_id].append(document_id)
    return dict(entities_per_document), dict(documents_per_entity)


def windows(token_ids, window_size, stride, padding_id=None):
    """Sliding windows over a document's in-vocabulary token ids."""
    out = []
    n = len(token_ids)
    for start in range(0, n, stride):
        piece = token_ids[start:start + window_size]
        if len(piece) == window_size:
            out.append(tuple(piece))
        elif piece and padding_id is not None:
            out.append(tuple(piece) + (padding_id,) * (window_size - len(piece)))
        if start + window_size >= n:
            break
    return out


def to_arrays(instances, window_size, class_mapping, instance_dtype, shuffle):
    """[(doc_id, window, label dict)] -> (x (N, n) ids, y csr (N, V_e) float32)."""
    if shuffle:
        np.random.shuffle(instances)
    n = len(instances)
    x = np.array([w for _, w, _ in instances], dtype=instance_dtype).reshape(n, window_size)
    rows, cols, vals = [], [], []
    for i, (_, _, label) in enumerate(instances):
        for col, mass in sorted((class_mapping[e], m) for e, m in label.items()):
            rows.append(i)
            cols.append(col)
            vals.append(mass)
    y = sparse.csr_matrix((np.array(vals, dtype=np.float32), (rows, cols)),
                          shape=(n, len(class_mapping)))
    return x, y


def prepare(args):
    """The whole pipeline; ``args`` = namespace of bin/prepare.py's flags."""
    from sklearn.model_selection import train_test_split

    np.random.seed(args.seed)

    ignore = set(MARKUP_TOKENS)
    if args.remove_stopwords == 'nltk':
        logging.info('Using the built-in English stop list (nltk is not available offline).')
        ignore |= BUILTIN_STOPWORDS
    elif args.remove_stopwords != 'none':
        with open(args.remove_stopwords, 'r') as f:
            ignore |= set(filter(len, (line.strip().lower() for line in f)))

    if args.overlapping and args.stride is not None:
        raise ValueError('Option --overlapping passed concurrently with --stride.')
    stride = 1 if args.overlapping else (args.stride or args.window_size)
    args.stride = stride
    logging.info('Generating instances with stride %d.', stride)

    documents = list(iter_trec_documents(args.document_paths, args.encoding))
    words, tokens = extract_vocabulary(
        documents, min_count=args.vocabulary_min_count, max_size=args.vocabulary_max_size,
        min_word_size=args.vocabulary_min_word_size, ignore=ignore,
        with_padding=not args.no_padding)
    padding_id = None if args.no_padding else words[PADDING_TOKEN].id

    with open(args.assoc_path, 'r') as f:
        entities_per_document, documents_per_entity = read_associations(
            f, known_documents=set(doc_id for doc_id, _ in documents))
    logging.info('Found %d unique entities.', len(documents_per_entity))

    instances_per_label = collections.defaultdict(list)
    instances_per_document = {}
    max_document_length = 0
    skipped = 0
    for doc_id, text in documents:
        entities = entities_per_document.get(doc_id)
        if not entities:
            skipped += 1
            continue
        ids = [words[t].id for t in tokenize(text) if t in words]
        doc_windows = windows(ids, args.window_size, stride, padding_id)
        if not doc_windows:
            logging.error('Document "%s" yielded zero instances.', doc_id)
            continue
        label = dict((e, 1.0 / len(entities)) for e in entities)
        instances_per_document[doc_id] = len(doc_windows)
        max_document_length = max(max_document_length, len(doc_windows))
        instances_per_label[tuple(sorted(entities))].extend(
            (doc_id, w, label) for w in doc_windows)
    logging.info('Observed %d documents of which %d are not associated with an entity.',
                 len(documents), skipped)

    total = sum(len(v) for v in instances_per_label.values())
    target = int(float(total) / max(1, len(instances_per_label))) if args.resample else 0

    instances, instances_per_entity = [], collections.defaultdict(int)
    for label_key in sorted(instances_per_label):
        pool = instances_per_label[label_key]
        if not pool:
            continue
        if args.resample:
            assert target > 0
            chosen = [pool[np.random.randint(len(pool))] for _ in range(target)]
        else:
            chosen = pool
        for entity_id in label_key:
            instances_per_entity[entity_id] += len(chosen)
        instances.extend(chosen)

    training, validation = train_test_split(instances, test_size=args.validation_set_ratio)
    logging.info('Processed %d instances; training=%d, validation=%d.', len(instances),
                 len(training), len(validation))

    entity_indices, entity_indices_inv = {}, {}
    for entity_id in sorted(instances_per_entity):
        if instances_per_entity[entity_id]:
            entity_indices_inv[len(entity_indices)] = entity_id
            entity_indices[entity_id] = len(entity_indices)
    logging.info('Retained %d entities after instance creation.', len(entity_indices))

    with open(args.meta_output, 'wb') as f:
        for obj in (args, words, tokens, entity_indices_inv, documents_per_entity):
            pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)

    instance_dtype = np.min_scalar_type(max(1, len(words) - 1))
    data = {}
    data['x_train'], y_train = to_arrays(training, args.window_size, entity_indices,
                                         instance_dtype, not args.no_shuffle)
    if not args.no_instance_weights:
        data['w_train'] = np.array(
            [float(max_document_length) / instances_per_document[doc_id] for doc_id, _, _ in training],
            dtype=np.float32)
    data['x_validate'], y_validate = to_arrays(validation, args.window_size, entity_indices,
                                               instance_dtype, not args.no_shuffle)
    # sparse matrices ride in 0-d object arrays (np.savez semantics of the reference)
    for key, value in (('y_train', y_train), ('y_validate', y_validate)):
        box = np.empty((), dtype=object)
        box[()] = value
        data[key] = box
    with open(args.data_output, 'wb') as f:
        np.savez(f, **data)
    logging.info('Saved data sets.')
    return data, (words, tokens, entity_indices_inv)
