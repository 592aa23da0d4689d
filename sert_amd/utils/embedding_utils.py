"""word2vec binary reader (cvangysel.embedding_utils.load_binary_representations,
call site bin/train.py:131-151; the upstream helper is an un-vendored submodule, so the
format below is the public word2vec one: an ASCII header "<count> <dim>\\n", then per word
its UTF-8 bytes, one space, <dim> little-endian float32 values and an optional newline)."""
import numpy as np


def _wanted_words(vocabulary):
    """The caller passes whatever names its vocabulary: a set / list of words, a word -> entry
    mapping, or the id -> word mapping of the meta file (bin/train.py:135 passes `tokens`)."""
    if vocabulary is None:
        return None
    if isinstance(vocabulary, dict):
        names = set(k for k in vocabulary.keys() if isinstance(k, str))
        names.update(v for v in vocabulary.values() if isinstance(v, str))
        return names
    return set(vocabulary)


def load_binary_representations(path, vocabulary=None):
    """Yield (word, vector float32) from a word2vec ``.bin`` file.  When `vocabulary` is given,
    words outside it are skipped (compared case-insensitively: the caller looks vectors up by
    ``word.lower()``, bin/train.py:143-145)."""
    wanted = _wanted_words(vocabulary)
    if wanted is not None:
        wanted = wanted | set(w.lower() for w in wanted)
    with open(path, 'rb') as f:
        header = f.readline().split()
        num_words, dim = int(header[0]), int(header[1])
        nbytes = 4 * dim
        for _ in range(num_words):
            chars = []
            while True:
                ch = f.read(1)
                if ch == b' ' or ch == b'':
                    break
                if ch != b'\n':
                    chars.append(ch)
            word = b''.join(chars).decode('utf-8', 'replace')
            raw = f.read(nbytes)
            if len(raw) != nbytes:
                raise ValueError('%s: truncated vector for word %r' % (path, word))
            if wanted is None or word in wanted:
                yield word, np.frombuffer(raw, dtype='<f4').astype(np.float32)


def save_binary_representations(path, words, vectors):
    """Write the same format (tests, and exporting trained word tables)."""
    vectors = np.ascontiguousarray(vectors, dtype='<f4')
    assert vectors.ndim == 2 and len(words) == vectors.shape[0]
    with open(path, 'wb') as f:
        f.write(('%d %d\n' % vectors.shape).encode('ascii'))
        for word, vec in zip(words, vectors):
            f.write(word.encode('utf-8') + b' ' + vec.tobytes() + b'\n')
