"""Drop-in for ``sert.math_utils``: ``entropy(pk, ..., normalize=False)`` --
scipy's Shannon entropy, optionally divided by the entropy of the uniform
distribution over the same number of classes (so the result lies in [0, 1]).
Only the ``_debug`` side file of bin/query.py uses it."""
import math

import numpy as np
import scipy.stats


def entropy(pk, *args, **kwargs):
    normalize = bool(kwargs.pop('normalize', False))
    value = scipy.stats.entropy(pk, *args, **kwargs)
    if not normalize:
        return value
    base = kwargs.get('base')
    uniform_entropy = math.log(np.size(pk)) / (math.log(base) if base else 1.0)
    return value / uniform_entropy
