"""``sert`` -- the reference's import name, bound to the MI355X engine.

The reference's callers do ``from sert import inference, math_utils, models``
(bin/query.py:6, bin/train.py:6 of cvangysel/SERT).  With this repository on the path
those lines import the HIP-backed drop-ins of ``sert_amd`` unchanged: same class names,
constructor arguments, methods, return values and exceptions (sert_amd/models.py,
sert_amd/inference.py, sert_amd/math_utils.py).
"""
import sys

from sert_amd import inference, math_utils, models

for _name, _module in (('inference', inference), ('math_utils', math_utils), ('models', models)):
    sys.modules[__name__ + '.' + _name] = _module

__all__ = ['inference', 'math_utils', 'models']
