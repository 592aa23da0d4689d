"""CPU: the ONE line bench.py prints stays under the 8 kB the driver keeps of stdout (round 4's 27 kB line did not parse),
whatever the full record holds, and carries the contract's keys."""
import copy
import json
import os
import re

import bench
from tests import util as U

ROOT = U.ROOT

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


def _latest_kernel_summaries():
    """(kind, path) of the newest committed rocprofv3 kernel summary per profiled configuration."""
    import glob
    out = []
    for tag, kind in (('vs_c2', 'vectorspace'), ('c4', 'vectorspace'), ('ll_c2', 'loglinear')):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*_%s_kernels.txt' % tag)))
        if files:
            out.append((kind, files[-1]))
    return out


def test_every_profiled_kernel_maps_to_a_group():
    """bench.KERNELS_OF_GROUP names the HIP kernels of every timing group; traffic_by_group looks the PMC bytes of a group up
    by those names.  Round 5 renamed the loglinear per-word dZ sums' kernel (segsum_rows_plus_ll) and the map kept the old
    name: `loglinear.roofline.frac_counter` went out as 0.0.  Every kernel holding more than 1 % of the kernel time of the
    newest committed summaries (profiles/rNN*_{vs_c2,c4,ll_c2}_kernels.txt) must map to a group of its model kind, or be one
    of the micro-benchmarks' / the runtime's kernels."""
    summaries = _latest_kernel_summaries()
    assert len(summaries) == 3, summaries
    for kind, path in summaries:
        unknown, seen = [], 0
        for line in open(path).read().splitlines()[1:]:
            if line.startswith('total kernel time') or not line.strip():
                continue
            m = re.match(r'^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s', line)
            assert m, line
            name, pct = m.group(1), float(m.group(7))
            seen += 1
            if pct > 1.0 and bench.group_of_kernel(kind, name) is None:
                unknown.append((name, pct))
        assert seen > 5 and not unknown, (path, unknown)
    # the one that went stale
    assert bench.group_of_kernel('loglinear', 'segsum_rows_plus_ll(float const*, int const*') == 'per_word_dz_sums'
    assert bench.group_of_kernel('vectorspace', 'dense_update_skip<true, 64, 2>(float*') == 'optimizer_word_table'
    assert bench.group_of_kernel('vectorspace', 'some_new_kernel(float*)') is None
    assert bench.group_of_kernel('vectorspace', 'mb_stream_copy(HIP_vector_type') == ''


def test_traffic_lookup_finds_the_dominant_loglinear_kernel_in_a_committed_counter_pass():
    """The replay of round 5's failure: traffic_by_group over the committed counter pass of the loglinear record
    (profiles/r05m_ll_c2_pmc.json, the rocpd_pmc per-kernel table the bench builds of its own --pmc passes) must find the
    per-word dZ sums' kernel with its 1.98 GB per big launch -- not an empty record under a stale name."""
    per_kernel = json.load(open(os.path.join(ROOT, 'profiles', 'r05m_ll_c2_pmc.json')))
    tbg = bench.traffic_by_group(per_kernel, {'per_word_dz_sums': 287.0, 'loss': 198.0, 'gemm_fwd': 90.0}, 'loglinear')
    assert tbg['per_word_dz_sums']['hip_kernel'].startswith('segsum_rows_plus_ll'), tbg['per_word_dz_sums']
    assert tbg['per_word_dz_sums']['hbm_bytes'] > 1.5e9, tbg['per_word_dz_sums']
    assert tbg['loss']['hip_kernel'].startswith('ll_row_wave') and tbg['gemm_fwd']['hip_kernel'].startswith('gemm_x3<false, false, 1')
