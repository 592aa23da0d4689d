"""CPU: the ONE line bench.py prints stays under the 8 kB the driver keeps of stdout (round 4's 27 kB line did not parse),
whatever the full record holds, and carries the contract's keys."""
import copy
import json
import os

import bench
from tests import util as U

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _full_record():
    with open(os.path.join(U.ROOT, 'profiles', 'r04x_bench_steps20.json')) as f:
        return json.load(f)


def test_line_of_a_real_full_record_is_short_and_complete():
    full = _full_record()
    assert len(json.dumps(full)) > 20000          # (the record that did not parse in round 4)
    rec = bench.compact_record(full, sidecar='gpurun_out/bench_full.json')
    line = json.dumps(rec)
    assert len(line) < bench.LINE_LIMIT
    assert '\n' not in line
    for k in CONTRACT:
        assert k in rec, k
    assert rec['value'] == float('%.6g' % full['value'])
    assert rec['config']['workload'].startswith('C2 LSE')
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_counter'):
        assert k in rec['roofline'], k
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in rec['cpu_baseline'], k
    for sub in ('loglinear', 'c4', 'query', 'lse_full_softmax', 'seeds', 'gemm_fp32_mfma_path'):
        assert sub in rec, sub
    assert rec['c4']['roofline']['frac'] is not None


def test_line_stays_short_when_the_record_grows():
    full = _full_record()
    big = copy.deepcopy(full)
    big['kernels'].update({'made_up_group_%d' % i: {'us': 1.0 + i, 'bound': 'latency', 'note': 'x' * 200} for i in range(200)})
    big['cpu_baseline']['sample'] = 'y' * 20000
    big['config']['workload'] = full['config']['workload']
    big['seeds'].update({'seed_%d' % i: {'value': 1.0, 'ms_per_step': 1.0} for i in range(3, 400)})
    big['device'] = 'z' * 5000
    rec = bench.compact_record(big)
    assert len(json.dumps(rec)) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in rec, k
    assert rec['roofline']['frac'] == full['roofline']['frac']


def test_line_without_optional_parts():
    full = _full_record()
    for k in ('loglinear', 'c4', 'query', 'lse_full_softmax', 'seeds', 'gemm_fp32_mfma_path', 'kernels'):
        full.pop(k, None)
    full['cpu_baseline'] = None
    full['roofline']['traffic'] = None
    rec = bench.compact_record(full)
    assert rec['cpu_baseline'] is None and rec['roofline']['traffic'] is None
    json.loads(json.dumps(rec))
