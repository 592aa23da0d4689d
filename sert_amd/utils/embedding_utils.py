"""word2vec binary reader (cvangysel.embedding_utils.load_binary_representations,
call site bin/train.py:131-151)."""
import numpy as np


def load_binary_representations(path, vocabulary=None):
    """Yield (word, vector float32) from a word2vec ``.bin`` file.  When
    `vocabulary` (a container of words) is given, other words are skipped."""
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
            vec = np.frombuffer(f.read(nbytes), dtype=np.float32)
            if vocabulary is None or word in vocabulary:
                yield word, vec.copy()
