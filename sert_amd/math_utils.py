"""Drop-in for ``sert.math_utils``: Shannon entropy, optionally normalised by
the maximum entropy log(num_classes) (sert/math_utils.py:5-25).  Only used for
the ``_debug`` side file of bin/query.py."""
import numpy as np
import scipy.stats


def entropy(pk, *args, **kwargs):
    normalize = kwargs.pop('normalize', False)

    e = scipy.stats.entropy(pk, *args, **kwargs)

    if normalize:
        maximum_entropy = np.log(np.size(pk))
        base = kwargs.get('base')
        if base:
            maximum_entropy /= np.log(base)
        e /= maximum_entropy

    return e
