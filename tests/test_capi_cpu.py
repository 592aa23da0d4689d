"""CPU: the C-ABI library loads, exports every symbol include/sert_hip.h
declares, agrees on the config struct, and the product path fails LOUDLY
without a GPU (no CPU fallback, no oracle import)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from sert_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header='sert_hip.h'):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sert_[a-z_0-9]+)\s*\(', src)))


def test_header_symbols_are_exported(hip_lib):
    declared = _declared_functions()
    debug = _declared_functions('sert_hip_debug.h')
    assert len(declared) >= 25
    missing = [s for s in declared + debug if not hasattr(hip_lib, s)]
    assert not missing, missing
    # the boundary header holds no test hook or micro-benchmark, the debug header nothing else
    assert not [s for s in declared if s.startswith(('sert_debug_', 'sert_bench_'))]
    assert all(s.startswith(('sert_debug_', 'sert_bench_')) for s in debug), debug
    # and the binding's list is the two headers' list
    assert sorted(_capi.EXPORTS) == sorted(declared + debug)


def test_product_host_code_uses_the_boundary_only():
    """The host-side product (models, inference, scoring, training, prepare, the CLIs) calls no test hook."""
    bad = []
    for base, names in (('sert_amd', ('models.py', 'inference.py', 'scoring.py', 'training.py', 'prepare.py', 'distributed.py')),
                        ('bin', ('train.py', 'query.py', 'prepare.py'))):
        for f in names:
            txt = open(os.path.join(ROOT, base, f)).read()
            if re.search(r'sert_debug_|sert_bench_|debug_gemm|bench_memory|bench_gemm|debug_word_index|debug_row_lists', txt):
                bad.append(f)
    assert not bad, bad


def test_config_struct_abi(hip_lib):
    """sert_create checks struct_size first: a wrong size yields the ABI message,
    the right size gets past it (and then fails on the missing device)."""
    cfg = _capi.SertConfig()
    cfg.struct_size = ctypes.sizeof(_capi.SertConfig) + 4
    h = ctypes.c_void_p()
    assert hip_lib.sert_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b'size mismatch' in hip_lib.sert_last_error()
    cfg.struct_size = ctypes.sizeof(_capi.SertConfig)
    cfg.kind = 7
    assert hip_lib.sert_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b'bad kind' in hip_lib.sert_last_error()


@pytest.mark.skipif(_capi.device_count() > 0, reason='needs a GPU-less host')
def test_fails_loudly_without_gpu(hip_lib):
    from sert_amd import models
    with pytest.raises(_capi.SertError):
        _capi.require_gpu()
    x = np.zeros((8, 2), dtype=np.uint8)
    y = np.zeros(8, dtype=np.int32)
    w = np.ones(8, dtype=np.float32)
    with pytest.raises(_capi.SertError):
        models.VectorSpaceLanguageModel(
            batch_size=4, window_size=2, num_negative_samples=2,
            representations_init=np.zeros((10, 4), np.float32),
            entity_representations_init=np.zeros((3, 4), np.float32),
            regularization_lambda=0.01, training_set=(x, y, w),
            validation_set=(x[:0], y[:0]))
    with pytest.raises(_capi.SertError):
        _capi.score_topk(np.ones((4, 2), np.float32), np.ones((1, 2), np.float32), 2)


def test_product_never_imports_the_oracle():
    """Nothing under sert_amd/ or bin/ may import, call or execute oracle/."""
    bad = []
    for base in ('sert_amd', 'bin'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(('.py', '.h', '.hip')):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'sert_oracle' in txt:
                        bad.append(os.path.join(dirpath, f))
                    # nor the test-only librccl stand-in (tests/rccl_stub): the product dlopen()s librccl by
                    # its bare name and knows nothing else about it
                    if 'rccl_stub' in txt:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    code = ("import sys; sys.path.insert(0, %r); import sert_amd.models, sert_amd.inference, "
            "sert_amd.scoring; assert not any(m.startswith('oracle') for m in sys.modules)" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])


def test_reference_import_name_binds_the_engine_and_needs_no_torch():
    """`from sert import inference, math_utils, models` (bin/query.py:6, bin/train.py:6 of the
    reference) resolves to the HIP-backed modules, with the reference's class names in place, and
    importing the product (distributed launcher included) pulls in neither PyTorch nor the oracle."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from sert import inference, math_utils, models\n"
        "import sert.models, sert_amd.distributed, sert_amd.training, sert_amd.scoring\n"
        "assert sert.models is models and models.__name__ == 'sert_amd.models'\n"
        "for name in ('LanguageModel', 'VectorSpaceLanguageModel', 'ModelBase', 'ModelInterface'):\n"
        "    assert hasattr(models, name), name\n"
        "assert hasattr(inference, 'create') and hasattr(math_utils, 'entropy')\n"
        "assert 'torch' not in sys.modules, 'the product imported PyTorch'\n"
        "assert not any(m.startswith('oracle') for m in sys.modules)\n" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])
