"""-m gpu: the HIP path (through the C ABI) against the CPU oracle.

Tolerances (stated, fp32): per-step loss rel 1e-5; activations rel 1e-5;
gradients rel 1e-4 (fp32 reassociation in atomics / MFMA / split-K);
parameters after k steps rel 1e-4.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import sert_oracle as O
from sert_amd import _capi as C
from tests import util as U

pytestmark = pytest.mark.gpu

LOSS_TOL, ACT_TOL, GRAD_TOL, PARAM_TOL = 1e-5, 2e-5, 1e-4, 1e-4



# The GEMM variants that lost their A/B (csrc/variants/) are not part of the product library: their
# test ids exist only when the suite runs against a library built with them
# (tools/build_variant.sh variants -DSERT_VARIANTS; SERT_LIB=sert_amd/variants/libsert_variants.so pytest ...).
VARIANTS_BUILD = 'variants' in os.path.basename(os.environ.get('SERT_LIB', ''))


@pytest.mark.parametrize('dims', [
    dict(B=64, n=5, z=4, Vw=500, Ve=37, dw=32, de=48),      # vector path, NPL=1
    dict(B=96, n=3, z=7, Vw=200, Ve=11, dw=30, de=70),      # scalar path, NPL=2, ragged tiles
    dict(B=256, n=10, z=10, Vw=3000, Ve=1000, dw=128, de=128),  # C2-shaped
    dict(B=130, n=4, z=10, Vw=70000, Ve=300, dw=300, de=128),   # uint32 ids, d=300
    dict(B=2100, n=2, z=5, Vw=300, Ve=5000, dw=16, de=32),      # 13 key bits: 2 sort passes, >1 tile/chunk carries
    dict(B=300, n=2, z=20, Vw=50, Ve=3, dw=8, de=300),          # heavy duplicate entities: long carry chains
    dict(B=5000, n=2, z=10, Vw=300, Ve=2, dw=16, de=256),       # two entities, d_e = 256: LDS path with 2 float4 per lane, queue drains mid-scan
    dict(B=4100, n=2, z=3, Vw=300, Ve=2048, dw=16, de=64),      # largest vocabulary of the LDS path, ragged last row group
    dict(B=1100, n=3, z=4, Vw=2000, Ve=50, dw=128, de=128),     # strip GEMMs (gemm_strip.h), ragged last strip
    dict(B=1030, n=2, z=3, Vw=500, Ve=40, dw=96, de=64),        # strip GEMMs with idle waves (N = 64 / 96), K = 96 / 64
    dict(B=20000, n=5, z=3, Vw=300, Ve=40, dw=32, de=16),       # three dense heavy words (> 4096 occurrences each: segsum_heavy), the rest through the tree
    dict(B=6000, n=5, z=2, Vw=300, Ve=30, dw=300, de=32),       # one dense heavy word at d_w = 300 (64 lanes x 2 chunks)
])
@pytest.mark.parametrize('egrad', ['default', 'sorted'] + (['strip_gemm', 'roles_gemm'] if VARIANTS_BUILD else []))
def test_vectorspace_steps(hip_lib, dims, egrad, monkeypatch):
    # entity gradient: V_e <= 2048 takes the sort-free bucket + register-accumulator path by default;
    # 'sorted' forces the counting-sort + chunked-reduce path every vocabulary size can take;
    # 'strip_gemm' switches the opt-in strip-streaming projection GEMMs on (gemm_strip.h)
    if egrad == 'sorted':
        monkeypatch.setenv('SERT_EGRAD_SORT', '1')
    if dims['B'] >= 6000:
        monkeypatch.setenv('SERT_DENSE_HEAVY', '1')    # (opt-in for vectorspace: the dense heavy-word pass)
    if egrad in ('strip_gemm', 'roles_gemm'):
        if dims['B'] < 1024:
            pytest.skip('strip GEMMs take M >= 1024 only')
        # 1: ping-pong strips; 2: role-specialised waves (compute / loader / epilogue)
        monkeypatch.setenv('SERT_STRIP_GEMM', '1' if egrad == 'strip_gemm' else '2')
    B, n, z = dims['B'], dims['n'], dims['z']
    steps = 3
    p = U.make_vs_problem(0, B * steps, n, z, dims['Vw'], dims['Ve'], dims['dw'], dims['de'],
                          zipf=True)
    lam = 0.01
    eng = U.vs_engine(p, B, n, z, lam)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], lam)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        neg = p['rng'].randint(0, dims['Ve'], size=(B, z)).astype(np.int64)
        loss_ref, grads_ref, f = ora.loss_and_grads(p['X'][sl], p['y'][sl], p['w'][sl], neg)
        ora.opt.update(ora.params(), grads_ref)
        loss = eng.train_batch(s, neg)
        assert abs(loss - loss_ref) <= LOSS_TOL * abs(loss_ref), (s, loss, loss_ref)
        assert U.rel_err(eng.get_tensor(C.T_ACT_H, (B, dims['dw'])), f['h']) < ACT_TOL
        assert U.rel_err(eng.get_tensor(C.T_ACT_T, (B, dims['de'])), f['t']) < ACT_TOL
        assert U.rel_err(eng.get_tensor(C.T_ACT_DA, (B, dims['de'])), f['da']) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_ACT_DH, (B, dims['dw'])), f['dh']) < GRAD_TOL
        dRe, dRw, dW, db = grads_ref
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RE), dRe.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RW), dRw.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_W), dW.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_B), db.ravel()) < GRAD_TOL
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_RE), ora.R_e.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_W), ora.W.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_B), ora.b.ravel()) < PARAM_TOL
    # eval: unweighted, unregularised, no update
    neg = p['rng'].randint(0, dims['Ve'], size=(B, z)).astype(np.int64)
    before = eng.get_tensor(C.T_RW).copy()
    ev = eng.eval_batch(C.SPLIT_TRAIN, 1, neg)
    ev_ref = ora.eval_loss(p['X'][B:2 * B], p['y'][B:2 * B], neg)
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    assert np.array_equal(before, eng.get_tensor(C.T_RW))
    eng.close()


@pytest.mark.parametrize('dims,groups,dense', [
    (dict(B=1000, n=10, Vw=3000, dw=128), 8, False),     # ragged row ranges (125 rows), one float4 chunk per lane
    (dict(B=1000, n=10, Vw=3000, dw=128), 24, False),    # three ranges per XCD list, 42-row ranges, the last one short
    (dict(B=777, n=4, Vw=60, dw=300), 16, False),        # d_w = 300: 32-lane groups x 3 column groups (2-D grid); few, heavy words
    (dict(B=900, n=3, Vw=5000, dw=256), 8, False),       # d_w = 256: 64-lane groups, four items per workgroup
    (dict(B=640, n=2, Vw=200, dw=16), 10, False),        # fewer row ranges than a multiple of eight
    (dict(B=20000, n=5, Vw=300, dw=32), 16, True),       # with the dense heavy-word pass: those words are absent from the lists
    (dict(B=9000, n=12, Vw=40, dw=64), 8, False),        # every word in every range, thousands of chunks: four tree levels
])
@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_SEG_GROUPS is read by a -DSERT_VARIANTS library only (measured slower)')
def test_word_gradient_row_grouped_tree(hip_lib, monkeypatch, dims, groups, dense):
    """The word-table gradient through the ROW-GROUPED tree (word_index.h: row_groups; what batches whose dh
    exceeds an XCD's L2 take -- C2, C4 -- forced here on small, ragged shapes): per (row range, word) items on
    eight XCD lists, words with a single item stored finally by level 0, the others through word-major partial
    rows and the upper levels.  Row by row against the float64 oracle, bit-identical run to run, and equal to the
    ungrouped tree up to fp32 reassociation."""
    monkeypatch.setenv('SERT_SEG_GROUPS', str(groups))
    if dense:
        monkeypatch.setenv('SERT_DENSE_HEAVY', '1')
    B, n, Vw, dw = dims['B'], dims['n'], dims['Vw'], dims['dw']
    z, Ve, de = 3, 20, 32
    p = U.make_vs_problem(5, B, n, z, Vw, Ve, dw, de, zipf=True)
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    _, g64, _ = o64.loss_and_grads(p['X'], p['y'], p['w'], neg)
    touched = np.unique(p['X'])
    got = []
    for run, g in enumerate((groups, groups, 1)):
        monkeypatch.setenv('SERT_SEG_GROUPS', str(g))
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        eng.train_batch(0, neg)
        got.append(eng.get_tensor(C.T_GRAD_RW, (Vw, dw)).copy())
        eng.close()
    assert np.array_equal(got[0], got[1])
    err, row = U.row_err(got[0], g64[1], rows=touched)
    assert err < 2e-5, (err, row)
    assert U.rel_err(got[0], got[2]) < 1e-5
    untouched = np.setdiff1d(np.arange(Vw), touched)
    if len(untouched):      # rows no token points to carry the L2 term only (keep_grads: dense, zeroed table)
        assert U.rel_err(got[0][untouched], g64[1][untouched]) < 1e-6


@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_GATHER_HOT (csrc/variants/kernels_gather_hot.h, measured 10 % slower) is read only by a library built with -DSERT_VARIANTS since round 6')
@pytest.mark.parametrize('dims', [
    dict(B=20000, n=5, Vw=300, dw=128),      # three hot words, the rest through L1 / L2
    dict(B=6100, n=8, Vw=3000, dw=300),      # 75 float4 per row, ragged last workgroup pass
])
def test_gather_with_hot_rows_in_lds_is_bit_identical(hip_lib, monkeypatch, dims):
    """SERT_GATHER_HOT=1 (opt-in, kernels_vs.h: vs_gather_mean_hot): the forward's gather takes the rows of the batch's dense
    heavy words from LDS instead of fetching them once per occurrence -- the same rows added in the same window order: h, the
    loss and the word-table gradient of a step are the plain kernel's bit for bit."""
    B, n, Vw, dw = dims['B'], dims['n'], dims['Vw'], dims['dw']
    z, Ve, de = 3, 20, 32
    p = U.make_vs_problem(17, B, n, z, Vw, Ve, dw, de, zipf=True)
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    got = []
    for hot in ('1', '0'):
        monkeypatch.setenv('SERT_GATHER_HOT', hot)
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        loss = eng.train_batch(0, neg)
        got.append((loss, eng.get_tensor(C.T_ACT_H, (B, dw)).copy(), eng.get_tensor(C.T_GRAD_RW, (Vw, dw)).copy()))
        eng.close()
    assert got[0][0] == got[1][0]
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    h64 = p['Rw'].astype(np.float64)[p['X'].astype(np.int64)].mean(axis=1)
    assert U.rel_err(got[0][1], h64) < 2e-6


@pytest.mark.parametrize('dims', [
    dict(B=20000, n=5, Vw=300, dw=128),      # C2's row width: a handful of words above 4096 occurrences, 157 row blocks (the last one ragged)
    dict(B=6100, n=8, Vw=3000, dw=300),      # d_w = 300: three 32-lane column groups (the extra workgroups' slab is blockIdx.y)
    dict(B=9000, n=12, Vw=20, dw=64, uniform=1),   # twenty words of ~5400 occurrences: sixteen dense, four stay in a three-level tree
    dict(B=5000, n=2, Vw=200000, dw=32, one=7),    # one word takes half the tokens, every other word occurs once or twice: NO level 1 (the combine alone)
])
def test_word_gradient_heavy_words_inside_the_tree_launches(hip_lib, monkeypatch, dims):
    """The dense heavy words of the vectorspace word gradient (word_index.h: kHeavyMax words above kHeavyMinCount
    occurrences) are summed by extra workgroups of the tree's own launches (kernels_seg.h: segsum_rows_plus -- the
    count-weighted stream beside level 0, its combine beside level 1; the default since round 5): row by row against the
    float64 oracle, bit-identical run to run, equal to the plain tree (SERT_DENSE_HEAVY=0) up to fp32 reassociation and
    -- against a variants build -- to the two launches in front of the tree (SERT_HEAVY_NO_FUSE=1)."""
    B, n, Vw, dw = dims['B'], dims['n'], dims['Vw'], dims['dw']
    z, Ve, de = 3, 20, 32
    p = U.make_vs_problem(13, B, n, z, Vw, Ve, dw, de, zipf=True)
    if 'uniform' in dims:
        p['X'] = np.random.RandomState(4).randint(0, Vw, size=p['X'].shape).astype(p['X'].dtype)
    if 'one' in dims:
        rs = np.random.RandomState(3)
        X = rs.randint(0, Vw, size=p['X'].shape).astype(p['X'].dtype)
        X[rs.rand(*X.shape) < 0.5] = dims['one']
        p['X'] = X
    counts = np.bincount(p['X'].ravel(), minlength=Vw)
    assert (counts > 4096).sum() >= 1, 'the shape has no heavy word'
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    _, g64, _ = o64.loss_and_grads(p['X'], p['y'], p['w'], neg)
    touched = np.unique(p['X'])
    runs = [{}, {}, {'SERT_DENSE_HEAVY': '0'}] + ([{'SERT_HEAVY_NO_FUSE': '1'}] if VARIANTS_BUILD else [])
    got = []
    for env in runs:
        monkeypatch.delenv('SERT_DENSE_HEAVY', raising=False)
        monkeypatch.delenv('SERT_HEAVY_NO_FUSE', raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        eng.train_batch(0, neg)
        got.append(eng.get_tensor(C.T_GRAD_RW, (Vw, dw)).copy())
        eng.close()
    assert np.array_equal(got[0], got[1])
    for g in got:
        err, row = U.row_err(g, g64[1], rows=touched)
        assert err < 2e-5, (err, row, int(counts[row]))
    heavy = np.argsort(-counts)[:16]
    assert not np.array_equal(got[0][heavy], got[2][heavy]), 'the dense pass did not run (same bits as the plain tree)'


@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_SEG_BUNDLE (csrc/variants/kernels_seg_bundled.h, measured slower) is read only by a library built with -DSERT_VARIANTS since round 6')
@pytest.mark.parametrize('dims', [
    dict(B=64, n=4, Vw=300, dw=16),          # mostly singletons: bundles of eight one-entry items
    dict(B=1000, n=10, Vw=5000, dw=128),     # Zipf: singletons, mid-size words and multi-chunk words in one batch
    dict(B=2000, n=6, Vw=50, dw=64),         # every word heavy: bundles of one 64-entry chunk item, three tree levels
    dict(B=777, n=5, Vw=4000, dw=300),       # 75 float4 per row: three column groups
    dict(B=3, n=2, Vw=7, dw=8),              # fewer items than one bundle
])
def test_word_gradient_bundled_level0(hip_lib, monkeypatch, dims):
    """Level 0 of the word-gradient tree in BUNDLES (kernels_seg.h: segsum_rows_bundled; word_index.h: bundle_off): up to
    eight consecutive short items per lane group, eight row loads in flight, every item still summed left to right --
    bit for bit the gradient of the one-item-per-lane-group kernel (the default; SERT_SEG_BUNDLE=1 at upload switches the
    bundles on: measured slower, kept as an opt-in), and row by row the float64 oracle's."""
    B, n, Vw, dw = dims['B'], dims['n'], dims['Vw'], dims['dw']
    z, Ve, de = 3, 20, 32
    p = U.make_vs_problem(9, B, n, z, Vw, Ve, dw, de, zipf=True)
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    got = []
    # (the tree alone: with bundles the dense heavy-word pass -- on by default since round 5 -- runs as two launches of its own,
    #  without them inside the tree's launches, over different row blocks: a different association for those words)
    monkeypatch.setenv('SERT_DENSE_HEAVY', '0')
    for bundle in ('1', '0'):
        monkeypatch.setenv('SERT_SEG_BUNDLE', bundle)
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        eng.train_batch(0, neg)
        got.append(eng.get_tensor(C.T_GRAD_RW, (Vw, dw)).copy())
        eng.close()
    assert np.array_equal(got[0], got[1])
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    _, g64, _ = o64.loss_and_grads(p['X'], p['y'], p['w'], neg)
    err, row = U.row_err(got[0], g64[1], rows=np.unique(p['X']))
    assert err < 2e-5, (err, row)


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_device_scope_events_change_nothing(hip_lib, monkeypatch, kind):
    """The events that order a model's own streams against each other are created without the system-scope fence
    (hipEventDisableSystemFence; SERT_EVENT_FENCE=system, read at sert_create, restores HIP's default flags): every kernel
    ends with an agent-scope release, which is all a consumer on the same device needs.  Same losses, same parameters,
    bit for bit, through the multi-stream schedule (device-drawn negatives, hints: run-ahead, deferred entity update)."""
    B, n, z, Vw, Ve, d = 2048, 6, 5, 20000, 300, 64
    if kind == 'vectorspace':
        p = U.make_vs_problem(61, B * 4, n, z, Vw, Ve, d, d, zipf=True)
        mk = lambda: U.vs_engine(p, B, n, z, 0.01, keep_grads=0)
    else:
        p = U.make_ll_problem(61, B * 4, n, Vw, Ve, d, 'int')
        mk = lambda: U.ll_engine(p, B, n, 0.01, keep_grads=0)
    outs = []
    for fence in ('system', None):
        if fence:
            monkeypatch.setenv('SERT_EVENT_FENCE', fence)
        else:
            monkeypatch.delenv('SERT_EVENT_FENCE', raising=False)
        eng = mk()
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for s in range(12):
            eng.hint_next_batch((s + 1) % 4 if s < 11 else None)
            losses.append(eng.train_batch(s % 4))
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_W).copy(), eng.get_tensor(C.T_B).copy()))
        eng.close()
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_PROJ_FUSED (csrc/variants/kernels_proj.h, measured slower) is read only by a library built with -DSERT_VARIANTS since round 6')
@pytest.mark.parametrize('dims', [
    dict(B=20000, n=10, Vw=5000, dw=128, de=128, exact=True),    # the unfused projection runs gemm_x3 here: the same bits
    dict(B=16500, n=3, Vw=900, dw=64, de=96, exact=True),        # K = 64 (four k steps), 96 of the 128 tile columns
    dict(B=65, n=10, Vw=300, dw=128, de=128, exact=False),       # one full tile + one row; unfused: the fp32 MFMA kernel
    dict(B=100, n=4, Vw=200, dw=32, de=20, exact=False),
    dict(B=1, n=1, Vw=5, dw=16, de=4, exact=False),
    dict(B=200, n=12, Vw=1000, dw=112, de=128, exact=False),     # window > 10: two gather trips
])
def test_fused_projection_equals_the_two_launches(hip_lib, monkeypatch, dims):
    """kernels_proj.h: gather + mean-pool + tanh projection in one persistent, software-pipelined launch (opt-in,
    SERT_PROJ_FUSED=1 at sert_create, for d_w, d_e <= 128 and windows <= 10: measured slower than the two launches at C2)
    against the two launches it replaces: h bit for bit always; t bit for bit where the
    unfused projection runs the bf16-pipe kernel with the same term order (gemm_x3.h), to 2e-6 where it runs the fp32 MFMA
    kernel; and both against the float64 oracle."""
    B, n, Vw, dw, de = dims['B'], dims['n'], dims['Vw'], dims['dw'], dims['de']
    z, Ve = 3, 20
    p = U.make_vs_problem(13, B, n, z, Vw, Ve, dw, de, zipf=True)
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    got = []
    for fused in ('1', '0'):
        monkeypatch.setenv('SERT_PROJ_FUSED', fused)
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        loss = eng.train_batch(0, neg)
        got.append((eng.get_tensor(C.T_ACT_H, (B, dw)).copy(), eng.get_tensor(C.T_ACT_T, (B, de)).copy(), loss,
                    eng.get_tensor(C.T_RW).copy()))
        eng.close()
    assert np.array_equal(got[0][0], got[1][0])
    if dims['exact']:
        assert np.array_equal(got[0][1], got[1][1])
        assert got[0][2] == got[1][2] and np.array_equal(got[0][3], got[1][3])
    else:
        assert np.abs(got[0][1] - got[1][1]).max() < 2e-6
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    f = o64.forward(p['X'], p['y'], neg)
    assert np.abs(got[0][0] - f['h']).max() < 1e-6 * max(1.0, np.abs(f['h']).max())
    assert np.abs(got[0][1] - f['t']).max() < 2e-6


@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_EGRAD_RANGES (csrc/variants/kernels_egrad_ranges.h, the step no faster) is read only by a library built with -DSERT_VARIANTS since round 6')
@pytest.mark.parametrize('dims', [
    dict(B=512, n=4, z=10, Vw=300, Ve=32768, dw=32, de=128),        # the product-search table: 256 ranges, ~22 pairs each
    dict(B=300, n=3, z=5, Vw=200, Ve=2049, dw=16, de=36),           # just above the LDS path's 2048; last range of one entity
    dict(B=2000, n=3, z=7, Vw=200, Ve=5000, dw=16, de=300),         # three column passes per row (75 float4 pieces)
    dict(B=3000, n=2, z=3, Vw=50, Ve=4000, dw=8, de=16, skew=True), # every label in ONE range: 3000 pairs of one entity + the rest
    dict(B=9000, n=2, z=0, Vw=50, Ve=3000, dw=8, de=8, skew=True),  # z = 0, 9000 pairs in one range: beyond the list, the slow walk
])
def test_entity_gradient_of_few_pairs_over_a_mid_size_table(hip_lib, monkeypatch, dims):
    """csrc/variants/kernels_egrad_ranges.h: egrad_ranges -- few (pair, entity) keys over a table above the LDS path's 2048 entities (the reference's
    product-search regime): one launch, one workgroup per range of 32 entities (scan, LDS list, bitonic sort, one chain per
    entity in pair order) instead of counting sort + chunked reduce + fix-up.  Row by row against the float64 oracle,
    bit-identical run to run, equal to the sorted path (the default: the range kernel is an opt-in, SERT_EGRAD_RANGES=1 --
    twice as fast alone, no faster as a step) up to fp32 reassociation; a range with more pairs than its list holds takes the
    in-order walk."""
    B, n, z, Vw, Ve, dw, de = (dims[k] for k in ('B', 'n', 'z', 'Vw', 'Ve', 'dw', 'de'))
    p = U.make_vs_problem(21, B, n, max(z, 1), Vw, Ve, dw, de, zipf=True)
    rng = np.random.RandomState(3)
    if dims.get('skew'):
        p['y'][:] = 1234                              # one entity takes every label
    neg = rng.randint(0, Ve, size=(B, z)).astype(np.int64) if z else np.zeros((B, 0), np.int64)
    got = []
    for mode in ('ranges', 'ranges', 'sorted'):
        monkeypatch.setenv('SERT_EGRAD_RANGES', '0' if mode == 'sorted' else '1')      # (opt-in, read at sert_create)
        eng = U.vs_engine(p, B, n, z, 0.01)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        eng.train_batch(0, neg if z else None)
        got.append(eng.get_tensor(C.T_GRAD_RE, (Ve, de)).copy())
        eng.close()
    assert np.array_equal(got[0], got[1])
    assert U.rel_err(got[0], got[2]) < 1e-5
    o64 = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01, dtype=np.float64)
    _, g64, _ = o64.loss_and_grads(p['X'], p['y'], p['w'], neg)
    err, row = U.row_err(got[0], g64[0])
    assert err < 2e-5, (err, row)


def test_vectorspace_known_answers(hip_lib):
    """W=0,b=0 => loss = (1+z) log 2; all tokens equal => row grad = sum dh/n * n."""
    B, n, z, Vw, Ve, dw, de = 64, 4, 5, 50, 9, 16, 16
    p = U.make_vs_problem(3, B, n, z, Vw, Ve, dw, de, weights='ones')
    p['W'][:] = 0
    p['b'][:] = 0
    p['X'][:] = 7
    eng = U.vs_engine(p, B, n, z, 0.0)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    neg = p['rng'].randint(0, Ve, size=(B, z)).astype(np.int64)
    loss = eng.train_batch(0, neg)
    assert abs(loss - (1 + z) * np.log(2.0)) < 1e-5
    g = eng.get_tensor(C.T_GRAD_RW, (Vw, dw))
    dh = eng.get_tensor(C.T_ACT_DH, (B, dw))
    assert np.abs(g[np.arange(Vw) != 7]).max() == 0.0
    assert U.rel_err(g[7], dh.sum(axis=0)) < 1e-5 or np.abs(dh).max() == 0.0
    eng.close()


@pytest.mark.skipif(not VARIANTS_BUILD, reason='SERT_BWD_FUSED is read only by a library built with -DSERT_VARIANTS (measured equal: not in the product)')
@pytest.mark.parametrize('B', [1100, 4096 + 37])
def test_vectorspace_fused_backward(hip_lib, monkeypatch, B):
    """SERT_BWD_FUSED=1 (opt-in, gemm_bwd_fused.h): dh = da.W^T and dW = h^T.da (+ db) out of ONE launch, for
    d_w = d_e = 128 -- a ragged last 64-row strip, fewer strips than workgroups / several strips per
    workgroup; three steps against the oracle (loss, activations, every gradient, parameters)."""
    monkeypatch.setenv('SERT_BWD_FUSED', '1')
    test_vectorspace_steps(hip_lib, dict(B=B, n=3, z=4, Vw=2000, Ve=50, dw=128, de=128), 'default', monkeypatch)


def test_vectorspace_predict(hip_lib):
    p = U.make_vs_problem(5, 8, 3, 2, 40, 9, 24, 40)
    eng = U.vs_engine(p, 8, 3, 2, 0.0)
    avg = p['rng'].randn(13, 24).astype(np.float32)
    out = eng.predict_project(avg)
    ref = np.tanh(avg @ p['W'] + p['b'])
    assert U.rel_err(out, ref) < 1e-5
    eng.close()


@pytest.mark.parametrize('labels', ['int', 'csr'])
@pytest.mark.parametrize('dims', [
    dict(B=32, n=4, Vw=300, Ve=53, d=24),
    dict(B=64, n=5, Vw=10000, Ve=100, d=64),     # C1-shaped
    dict(B=40, n=3, Vw=500, Ve=1000, d=30),
    dict(B=33, n=6, Vw=400, Ve=400, d=20),        # one wave per row, two float4 chunks per lane (ll_row_wave<2>)
    dict(B=21, n=7, Vw=300, Ve=2048, d=12),       # ... eight chunks per lane: the largest row the wave kernel takes
    dict(B=8, n=10, Vw=300, Ve=3500, d=16),       # 154 KB slab: fused kernel
    dict(B=8, n=12, Vw=300, Ve=4000, d=16),       # > LDS: streaming path, one segment
    dict(B=5, n=5, Vw=300, Ve=12000, d=16),       # streaming path, 3 segments, 16-byte rows
    dict(B=4, n=4, Vw=200, Ve=9001, d=12),        # streaming path, ragged last segment, scalar rows
    dict(B=48, n=5, Vw=400, Ve=300, d=32),        # dW on 160-column tiles (V_e just above 256), K = distinct words
    dict(B=37, n=6, Vw=400, Ve=300, d=32),        # ... another count of distinct words (odd / even K on 16-byte loaders)
    dict(B=512, n=8, Vw=3000, Ve=4096, d=128),    # dW = G^T.dZ of > 2 GFLOP: dW, combine and W, b update on the side stream;
                                                  # an odd number of distinct words through the 16-byte k-major loaders
])
def test_loglinear_steps(hip_lib, dims, labels):
    B, n = dims['B'], dims['n']
    steps = 3
    p = U.make_ll_problem(1, B * steps, n, dims['Vw'], dims['Ve'], dims['d'], labels)
    lam = 0.01
    eng = U.ll_engine(p, B, n, lam)
    if labels == 'int':
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    else:
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], csr=p['y'], w=p['w'])
    ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], lam)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        loss_ref, grads_ref, f = ora.loss_and_grads(p['X'][sl], p['ydense'][sl], p['w'][sl])
        ora.opt.update(ora.params(), grads_ref)
        loss = eng.train_batch(s)
        assert abs(loss - loss_ref) <= LOSS_TOL * abs(loss_ref), (s, loss, loss_ref)
        dRw, dW, db = grads_ref
        assert U.rel_err(eng.get_tensor(C.T_GRAD_W), dW.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_B), db.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RW), dRw.ravel()) < GRAD_TOL
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_W), ora.W.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_B), ora.b.ravel()) < PARAM_TOL
    ev = eng.eval_batch(C.SPLIT_TRAIN, 0)
    ev_ref = ora.eval_loss(p['X'][:B], p['ydense'][:B])
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    # predict_fn: per-token distributions
    P = eng.predict_tokens(p['X'][:7])
    _, Pref = ora.token_distributions(p['X'][:7])
    assert U.rel_err(P, Pref) < 1e-5
    eng.close()


def test_loglinear_dense_heavy_words(hip_lib):
    """Words with more than 4096 occurrences in a batch: their rows of the per-word dZ sums come from the
    dense pass over dJ (segsum_heavy over V_e-wide rows + segsum_heavy_combine_ll), the other words' from
    the tree with the heavy words' items skipped.  Zipfian ids clipped to a small vocabulary put two such
    words into a batch of 9000 x 5 tokens; two steps against the oracle."""
    B, n, Vw, Ve, d, steps = 9000, 5, 300, 24, 16, 2
    rng = np.random.RandomState(3)
    p = U.make_ll_problem(3, B * steps, n, Vw, Ve, d, 'int')
    p['X'] = rng.permutation(Vw)[np.minimum(rng.zipf(1.1, size=(B * steps, n)) - 1, Vw - 1)].astype(p['X'].dtype)
    counts = np.bincount(p['X'][:B].ravel(), minlength=Vw)
    assert (counts > 4096).sum() >= 2
    lam = 0.01
    eng = U.ll_engine(p, B, n, lam)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], lam)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        loss_ref, grads_ref, _ = ora.loss_and_grads(p['X'][sl], p['ydense'][sl], p['w'][sl])
        ora.opt.update(ora.params(), grads_ref)
        loss = eng.train_batch(s)
        assert abs(loss - loss_ref) <= LOSS_TOL * abs(loss_ref), (s, loss, loss_ref)
        dRw, dW, db = grads_ref
        assert U.rel_err(eng.get_tensor(C.T_GRAD_W), dW.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_B), db.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RW), dRw.ravel()) < GRAD_TOL
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_W), ora.W.ravel()) < PARAM_TOL
    eng.close()


@pytest.mark.parametrize('V,d,Q,k', [(50, 16, 7, 10), (1000, 128, 33, 100), (5000, 300, 5, 100),
                                     (300, 64, 4, 300)])
def test_score_topk(hip_lib, V, d, Q, k):
    rng = np.random.RandomState(2)
    E = rng.randn(V, d).astype(np.float32)
    Pj = np.tanh(rng.randn(Q, d)).astype(np.float32)
    idx, val = C.score_topk(E, Pj, k)
    for q in range(Q):
        order, sc = O.vectorspace_rank(Pj[q].astype(np.float64), E.astype(np.float64), top=k)
        # identical ranking except where the fp64 score gap is below 1e-6
        mism = np.nonzero(idx[q] != order)[0]
        for r in mism:
            full = O.vectorspace_scores(Pj[q].astype(np.float64), E.astype(np.float64))
            assert abs(full[idx[q][r]] - sc[r]) < 1e-6
        assert np.abs(val[q] - sc).max() < 1e-6


def test_score_topk_ties(hip_lib):
    """Duplicate entity rows: ties resolve to the lowest index first."""
    rng = np.random.RandomState(4)
    E = rng.randn(20, 8).astype(np.float32)
    E = np.concatenate([E, E[:5]], axis=0)   # rows 20..24 duplicate 0..4
    Pj = E[:3].copy()                         # best match: itself and its duplicate
    idx, val = C.score_topk(E, Pj, 4)
    for q in range(3):
        assert idx[q][0] == q and idx[q][1] == 20 + q
        assert val[q][0] == val[q][1]


@pytest.mark.parametrize('V', [5000, 9000])
def test_score_topk_mass_ties(hip_lib, V):
    """Only 3 distinct entity directions: thousands of exactly equal scores.
    V=5000 keeps the threshold bin inside the LDS candidate list (fast path),
    V=9000 overflows it (4-pass fallback).  Either way: lowest index first."""
    rng = np.random.RandomState(6)
    dirs = rng.randn(3, 16).astype(np.float32)
    which = np.arange(V) % 3
    E = dirs[which]
    Pj = dirs[[2]] + 0.01 * dirs[[0]]
    k = 100
    idx, val = C.score_topk(E, Pj, k)
    expect = np.nonzero(which == 2)[0][:k]
    assert np.array_equal(idx[0], expect)
    assert np.all(val[0] == val[0][0])


def _check_topk_against_oracle(E, Pj, idx, val, k):
    E64, P64 = E.astype(np.float64), Pj.astype(np.float64)
    for q in range(Pj.shape[0]):
        order, sc = O.vectorspace_rank(P64[q], E64, top=k)
        full = None
        for r in np.nonzero(idx[q] != order)[0]:
            # identical ranking except where the fp64 score gap is below 1e-6
            if full is None:
                full = O.vectorspace_scores(P64[q], E64)
            assert abs(full[idx[q][r]] - sc[r]) < 1e-6
        assert np.abs(val[q] - sc).max() < 1e-6
        assert len(set(idx[q].tolist())) == k


@pytest.mark.parametrize('V,d,Q,k', [(40000, 16, 37, 10), (65536, 32, 130, 100), (50001, 64, 9, 1000),
                                     (33333, 300, 21, 100), (140001, 16, 5, 10)])
def test_score_topk_fused_filter_path(hip_lib, V, d, Q, k):
    """V >= 32768: sampled thresholds + GEMM with a filtering epilogue + selection from
    the candidate lists (the score matrix is never materialised); same contract."""
    rng = np.random.RandomState(11)
    E = rng.randn(V, d).astype(np.float32)
    Pj = np.tanh(rng.randn(Q, d)).astype(np.float32)
    idx, val = C.score_topk(E, Pj, k)
    _check_topk_against_oracle(E, Pj, idx, val, k)


def test_score_topk_fused_path_adversarial_rows(hip_lib):
    """Rows whose sampled threshold is useless are recomputed exactly:
    (a) a query aligned with thousands of identical entities (every sampled and
        unsampled copy ties: the candidate list overflows);
    (b) the best entities all sit at indices the stride-16 sample never sees
        (too few candidates would be impossible only by luck -- the flag catches it);
    (c) ordinary queries in the same call keep the fused result."""
    rng = np.random.RandomState(12)
    V, d, k = 40000, 16, 50
    E = rng.randn(V, d).astype(np.float32)
    hot = rng.randn(d).astype(np.float32)
    E[5000:15000] = hot                               # (a) 10 000 exact duplicates
    spike = rng.randn(d).astype(np.float32)
    off_sample = np.arange(20001, 20001 + 16 * 60, 16)   # (b) indices = 1 mod 16
    E[off_sample] = spike + 0.01 * rng.randn(len(off_sample), d).astype(np.float32)
    Pj = np.stack([hot, spike] + [np.tanh(rng.randn(d)).astype(np.float32) for _ in range(6)]).astype(np.float32)
    idx, val = C.score_topk(E, Pj, k)
    assert np.array_equal(idx[0], np.arange(5000, 5000 + k))       # ties -> lowest index first
    assert np.all(val[0] == val[0][0])
    assert set(idx[1].tolist()) <= set(off_sample.tolist())
    _check_topk_against_oracle(E, Pj[1:], idx[1:], val[1:], k)


def test_score_topk_fused_equals_materialised(hip_lib, monkeypatch):
    """The two scoring paths return identical indices and bit-identical scores."""
    rng = np.random.RandomState(13)
    V, d, Q, k = 70000, 32, 300, 100
    E = rng.randn(V, d).astype(np.float32)
    Pj = np.tanh(rng.randn(Q, d)).astype(np.float32)
    idx_f, val_f = C.score_topk(E, Pj, k)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from sert_amd import _capi as C;"
            "rng = np.random.RandomState(13); E = rng.randn(%d, %d).astype(np.float32);"
            "Pj = np.tanh(rng.randn(%d, %d)).astype(np.float32); idx, val = C.score_topk(E, Pj, %d);"
            "np.savez(sys.argv[1], idx=idx, val=val)" % (U.ROOT, V, d, Q, d, k))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'm.npz')
        env = dict(os.environ, SERT_SCORE_MATERIALISE='1')
        subprocess.run([sys.executable, '-c', code, out], check=True, env=env)
        ref = np.load(out)
        assert np.array_equal(idx_f, ref['idx'])
        assert np.array_equal(val_f, ref['val'])


def test_score_topk_fp32_filter_path(hip_lib):
    """SERT_SCORE_FP32=1: the fp32 filtering GEMM (gemm.h EPI_FILTER + topk_from_groups) that the
    scorer falls back to when the bf16 prefilter is demoted -- same contract."""
    rng = np.random.RandomState(31)
    V, d, Q, k = 45001, 32, 70, 100
    E = rng.randn(V, d).astype(np.float32)
    Pj = np.tanh(rng.randn(Q, d)).astype(np.float32)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from sert_amd import _capi as C;"
            "d = np.load(sys.argv[1]); idx, val = C.score_topk(d['E'], d['Pj'], %d);"
            "np.savez(sys.argv[2], idx=idx, val=val)" % (U.ROOT, k))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, 'in.npz'), os.path.join(tmp, 'out.npz')
        np.savez(inp, E=E, Pj=Pj)
        subprocess.run([sys.executable, '-c', code, inp, out], check=True,
                       env=dict(os.environ, SERT_SCORE_FP32='1'))
        r = np.load(out)
        _check_topk_against_oracle(E, Pj, r['idx'], r['val'], k)


def test_score_topk_bf16_prefilter_near_ties(hip_lib):
    """bf16 cannot order a cluster of near-duplicate entities (score gaps ~1e-5 << 2^-8): the
    prefilter may only decide what gets re-scored, the reported ranking is the fp32 one."""
    rng = np.random.RandomState(23)
    V, d, Q, k = 40000, 64, 24, 100
    E = rng.randn(V, d).astype(np.float32)
    u = rng.randn(d).astype(np.float32)
    cluster = rng.choice(V, 1500, replace=False)
    E[cluster] = u + 2e-3 * rng.randn(cluster.size, d).astype(np.float32)
    Pj = (u + 0.05 * rng.randn(Q, d)).astype(np.float32)
    idx, val = C.score_topk(E, Pj, k)
    assert np.isin(idx, cluster).all()          # the top 100 all come from the cluster
    _check_topk_against_oracle(E, Pj, idx, val, k)


def test_score_topk_bf16_prefilter_no_gap_falls_back(hip_lib):
    """Every entity within ~1e-2 of every other: no 2-delta gap between the k-th score and the
    filter threshold, so the bf16 path must hand the rows to the exact fp32 path."""
    rng = np.random.RandomState(29)
    V, d, Q, k = 36000, 32, 9, 50
    u = rng.randn(d).astype(np.float32)
    E = (u + 5e-3 * rng.randn(V, d)).astype(np.float32)
    Pj = (u + 0.05 * rng.randn(Q, d)).astype(np.float32)
    idx, val = C.score_topk(E, Pj, k)
    _check_topk_against_oracle(E, Pj, idx, val, k)


@pytest.mark.skipif(not VARIANTS_BUILD, reason='opt-in 256x256-tile scoring GEMM: not in the product build')
@pytest.mark.parametrize('mode', ['1', '2'] if VARIANTS_BUILD else ['-'])
def test_score_topk_big_tile_variant(hip_lib, mode):
    """SERT_SCORE_BIG_TILE=1|2 (variants/gemm_big.h, ragged M and N): fused == materialised, both == oracle."""
    rng = np.random.RandomState(17)
    V, d, Q, k = 70001, 32, 300, 100
    E = rng.randn(V, d).astype(np.float32)
    Pj = np.tanh(rng.randn(Q, d)).astype(np.float32)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from sert_amd import _capi as C;"
            "d = np.load(sys.argv[1]); idx, val = C.score_topk(d['E'], d['Pj'], %d);"
            "np.savez(sys.argv[2], idx=idx, val=val)" % (U.ROOT, k))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, 'in.npz')
        np.savez(inp, E=E, Pj=Pj)
        res = []
        for extra in ({}, {'SERT_SCORE_MATERIALISE': '1'}):
            out = os.path.join(tmp, 'o%d.npz' % len(res))
            subprocess.run([sys.executable, '-c', code, inp, out], check=True,
                           env=dict(os.environ, SERT_SCORE_BIG_TILE=mode, **extra))
            r = np.load(out)
            res.append((r['idx'], r['val']))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    _check_topk_against_oracle(E, Pj, res[0][0], res[0][1], k)


def test_device_sampler_uniform_and_rank_invariant(hip_lib):
    """The Philox sampler draws iid uniform ids; training with it is finite."""
    B, n, z, Vw, Ve, dw, de = 512, 4, 8, 100, 16, 16, 16
    p = U.make_vs_problem(7, B * 2, n, z, Vw, Ve, dw, de)
    eng = U.vs_engine(p, B, n, z, 0.01)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    losses = eng.train_batches([0, 1, 0, 1])
    assert np.all(np.isfinite(losses))
    l1 = eng.train_batch(0)
    assert np.isfinite(l1)
    eng.close()


@pytest.mark.parametrize('dims', [
    dict(B=1, n=1, z=1, Vw=5, Ve=2, dw=4, de=4),           # smallest legal everything
    dict(B=3, n=2, z=1, Vw=7, Ve=3, dw=8, de=12),          # B below every tile / group width
    dict(B=17, n=12, z=31, Vw=300, Ve=5, dw=64, de=64),    # z+1 = 32 candidates per row
    dict(B=33, n=3, z=2, Vw=256, Ve=300, dw=20, de=36),    # uint8 ids, id 255 used
])
def test_vectorspace_edge_shapes(hip_lib, dims):
    B, n, z = dims['B'], dims['n'], dims['z']
    p = U.make_vs_problem(41, 2 * B + 1, n, z, dims['Vw'], dims['Ve'], dims['dw'], dims['de'])
    p['X'][0, 0] = dims['Vw'] - 1
    eng = U.vs_engine(p, B, n, z, 0.01)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])   # 1 row of tail ignored
    ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
    for s in range(2):
        sl = slice(s * B, (s + 1) * B)
        neg = p['rng'].randint(0, dims['Ve'], size=(B, z)).astype(np.int64)
        ref = ora.train_step(p['X'][sl], p['y'][sl], p['w'][sl], neg)
        got = eng.train_batch(s, neg)
        assert abs(got - ref) <= LOSS_TOL * abs(ref)
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_RE), ora.R_e.ravel()) < PARAM_TOL
    with pytest.raises(C.SertError):
        eng.train_batch(3)            # beyond the data (an incomplete tail batch does not exist)
    eng.close()


def test_error_paths(hip_lib):
    p = U.make_vs_problem(42, 8, 2, 2, 10, 4, 4, 4)
    eng = U.vs_engine(p, 4, 2, 2, 0.0)
    with pytest.raises(C.SertError):
        eng.train_batch(0)                                   # nothing uploaded
    bad = p['X'].copy()
    bad[3, 1] = 10                                           # id == vocab_size
    with pytest.raises(C.SertError):
        eng.upload_dataset(C.SPLIT_TRAIN, bad, y_int=p['y'], w=p['w'])
    with pytest.raises(C.SertError):
        eng.set_tensor(C.T_RW, np.zeros(3, np.float32))      # wrong element count
    with pytest.raises(C.SertError):
        eng.predict_tokens(np.zeros((1, 2), np.uint8))       # loglinear-only entry point
    eng.close()
    with pytest.raises(C.SertError):
        C.score_topk(np.ones((4, 2), np.float32), np.ones((1, 2), np.float32), 5)   # k > V_e
    with pytest.raises(C.SertError):
        C.Engine(kind=C.KIND_VECTORSPACE, batch_size=4, global_batch_size=4, window_size=2,
                 vocab_size=10, num_entities=4, word_dim=4, entity_dim=4, num_negatives=2,
                 id_bytes=3, device=0)                       # id width must be 1, 2 or 4


def test_model_with_empty_validation_and_short_data(hip_lib):
    """validation set empty -> mean of no batches is nan (as np.mean([]) in the
    reference); fewer instances than one batch -> zero batches, no crash."""
    import warnings
    from sert_amd import models
    B, n, z, Vw, Ve, d = 16, 3, 2, 50, 6, 8
    p = U.make_vs_problem(43, B * 2 + 5, n, z, Vw, Ve, d, d)
    m = models.VectorSpaceLanguageModel(
        batch_size=B, window_size=n, num_negative_samples=z, representations_init=p['Rw'],
        entity_representations_init=p['Re'], regularization_lambda=0.01,
        training_set=(p['X'], p['y'], p['w']),
        validation_set=(np.zeros((0,), p['X'].dtype), np.zeros((0,), np.int32)))
    nb, mean = m.train()
    assert nb == 2 and np.isfinite(mean)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        vm, vs = m.validation_error()
    assert np.isnan(vm)
    # non-finite loss -> RuntimeError from the epoch loop (models.py:372-379)
    m._engine.set_tensor(C.T_W, np.full((d, d), np.nan, np.float32))
    with pytest.raises(RuntimeError):
        m.train()


@pytest.mark.parametrize('dims', [
    dict(B=64, n=5, Vw=500, Ve=37, dw=32, de=48),
    dict(B=96, n=3, Vw=200, Ve=1000, dw=30, de=68),
    dict(B=256, n=10, Vw=3000, Ve=1000, dw=128, de=128),     # C2-shaped
])
@pytest.mark.parametrize('tile', [None, 40])
def test_vectorspace_softmax_variant_steps(hip_lib, dims, tile, monkeypatch):
    """Additive full-softmax variant (SERT_KIND_VECTORSPACE_SOFTMAX) vs its oracle; tile = 40: the logits
    exist for 40 rows at a time (the path that keeps the C4 configuration's 26 GB logit matrix at 1.6 GB:
    per row tile logits, cross-entropy, dR_e += dZ^T.p through the accumulating GEMM epilogue, dp = dZ.R_e;
    ragged last tile)."""
    if tile:
        monkeypatch.setenv('SERT_FS_TILE_ROWS', str(tile))
    B, n = dims['B'], dims['n']
    steps = 3
    p = U.make_vs_problem(8, B * steps, n, 0, dims['Vw'], dims['Ve'], dims['dw'], dims['de'], zipf=True)
    eng = C.Engine(kind=C.KIND_VECTORSPACE_SOFTMAX, batch_size=B, global_batch_size=B, window_size=n,
                   vocab_size=dims['Vw'], num_entities=dims['Ve'], word_dim=dims['dw'],
                   entity_dim=dims['de'], num_negatives=0, id_bytes=p['X'].dtype.itemsize, device=0,
                   keep_grads=1, deterministic=1, lambda_=0.01, lr=1e-3, beta1=0.9, beta2=0.999,
                   eps=1e-8, seed=1)
    for which, a in ((C.T_RW, p['Rw']), (C.T_RE, p['Re']), (C.T_W, p['W']), (C.T_B, p['b'])):
        eng.set_tensor(which, a)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.VectorSpaceSoftmaxOracle(B, n, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        loss_ref, grads_ref, f = ora.loss_and_grads(p['X'][sl], p['y'][sl], p['w'][sl])
        ora.opt.update(ora.params(), grads_ref)
        loss = eng.train_batch(s)
        assert abs(loss - loss_ref) <= LOSS_TOL * abs(loss_ref), (s, loss, loss_ref)
        dRe, dRw, dW, db = grads_ref
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RE), dRe.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_RW), dRw.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_W), dW.ravel()) < GRAD_TOL
        assert U.rel_err(eng.get_tensor(C.T_GRAD_B), db.ravel()) < GRAD_TOL
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_RE), ora.R_e.ravel()) < PARAM_TOL
    ev = eng.eval_batch(C.SPLIT_TRAIN, 0)
    ev_ref = ora.eval_loss(p['X'][:B], p['y'][:B])
    assert abs(ev - ev_ref) <= LOSS_TOL * abs(ev_ref)
    eng.close()


@pytest.mark.parametrize('dw', [16, 10])
def test_untouched_word_rows_need_no_memset(hip_lib, dw):
    """keep_grads=0 (the product setting): the word-gradient table is neither zeroed nor
    read where no token of the batch points; the optimiser still applies the L2 / moment
    decay to those rows.  Must equal the keep_grads=1 run (dense, zeroed table) bit for
    bit.  dw=10 (not a multiple of 4) keeps the dense path in both runs."""
    B, n, z, Vw, Ve, de = 32, 3, 4, 5000, 12, 16      # 96 tokens per batch over 5000 words
    p = U.make_vs_problem(41, B * 4, n, z, Vw, Ve, dw, de)
    neg = p['rng'].randint(0, Ve, (B, z)).astype(np.int64)
    outs = []
    for keep in (1, 0):
        eng = U.vs_engine(p, B, n, z, 0.05, keep_grads=keep)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = [eng.train_batch(s % 4, neg) for s in range(6)]
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_RE).copy()))
        eng.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    # and the untouched rows did move (L2 + Adam on a zero data gradient)
    assert not np.array_equal(outs[1][1].reshape(Vw, dw)[-50:], p['Rw'][-50:])


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_lazy_word_table_update_is_bit_exact(hip_lib, kind):
    """The LAZY dense update of the word table (kernels_opt.h: dense_update_skip / dense_update_lazy; here a batch touches
    2 % of the rows): rows that neither the batch nor the ANNOUNCED next batch touches are neither read nor written (their
    share of sum(p^2) was left behind by the launch that wrote them), and brought forward in registers when they are needed
    -- by the same element update, so every parameter and optimiser moment must equal the dense run (keep_grads = 1: zeroed
    table, every row updated in memory every step) BIT FOR BIT and every loss to rounding, whatever the hints: right ones (rows stay behind for up to three updates), wrong ones
    (the forward finds its rows stale: flush), none (everything written), an evaluation and a tensor read in between
    (flush), a change of the step counter."""
    B, n, Vw, d, steps = 32, 3, 270000, 16, 14        # (4.3 M parameters: above the size below which the table stays dense)
    if kind == 'vectorspace':
        z, Ve = 4, 12
        p = U.make_vs_problem(43, B * 5, n, z, Vw, Ve, d, d)
        mk = lambda keep: U.vs_engine(p, B, n, z, 0.05, keep_grads=keep)
    else:
        p = U.make_ll_problem(43, B * 5, n, Vw, 24, d, 'int')
        mk = lambda keep: U.ll_engine(p, B, n, 0.05, keep_grads=keep)
    # per step: (batch, hint given before the step)  -- None: no hint
    plan = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0), (0, 3), (1, None), (2, 3), (3, 4), (4, 0), (0, 1), (1, 2), (2, 0), (0, None)]
    assert len(plan) == steps
    outs = []
    for keep in (1, 0):
        eng = mk(keep)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        rec = []
        for s, (b, hint) in enumerate(plan):
            if hint is not None:
                eng.hint_next_batch(hint)
            rec.append(eng.train_batch(b))
            if s == 4:
                rec.append(eng.eval_batch(C.SPLIT_TRAIN, 2) if kind == 'loglinear' else 0.0)   # (vs evals draw their own negatives)
            if s == 8:
                rec.append(float(eng.get_tensor(C.T_RW).sum()))
            if s == 10:
                eng.set_step(eng.get_step())          # (flushes; the counter itself is unchanged)
        outs.append((rec, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_STATE0_RW).copy(), eng.get_tensor(C.T_STATE1_RW).copy()))
        eng.close()
    # (losses: dense_update_skip adds the squares row by row -- a predicted row sum must be the one a reading launch forms --,
    #  the dense launches element by element: the same squares through another summation tree)
    assert all(abs(x - y) <= 2e-6 * abs(y) for x, y in zip(outs[0][0], outs[1][0])), (outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_next_batch_hint_changes_nothing_but_the_schedule(hip_lib, kind):
    """sert_hint_next_batch: the announced batch's forward + backward runs ahead of the host;
    it must not change any result -- with correct hints, with a WRONG hint (another batch is
    trained next), and with an evaluation in between (which reuses the buffers)."""
    B, n, z, Vw, Ve, dw, de = 64, 3, 4, 300, 12, 16, 16
    if kind == 'vectorspace':
        p = U.make_vs_problem(51, B * 5, n, z, Vw, Ve, dw, de)
        neg = p['rng'].randint(0, Ve, (B, z)).astype(np.int64)
    else:
        p = U.make_ll_problem(51, B * 5, n, Vw, Ve, dw, 'int')
    order = [3, 0, 4, 1, 2, 0]
    outs = []
    for mode in ('none', 'right', 'wrong'):
        if kind == 'vectorspace':
            eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0)
        else:
            eng = U.ll_engine(p, B, n, 0.01, keep_grads=0)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for pos, b in enumerate(order):
            nxt = order[pos + 1] if pos + 1 < len(order) else None
            if mode == 'right':
                eng.hint_next_batch(nxt)
            elif mode == 'wrong':
                eng.hint_next_batch((b + 2) % 5)
            # vectorspace: device-drawn negatives (explicit ones switch the run-ahead off)
            losses.append(eng.train_batch(b))
            if pos == 2:
                losses.append(eng.eval_batch(C.SPLIT_TRAIN, 1, neg) if kind == 'vectorspace'
                              else eng.eval_batch(C.SPLIT_TRAIN, 1))
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_W).copy()))
        eng.close()
    for other in outs[1:]:
        assert outs[0][0] == other[0]
        assert np.array_equal(outs[0][1], other[1])
        assert np.array_equal(outs[0][2], other[2])


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_parameters_after_100_steps(hip_lib, kind):
    """SURVEY 8(d) tolerance: parameters after 100 optimiser steps within rel. 1e-4 of the
    oracle (fresh negatives every step for vectorspace; 4 batches cycled)."""
    steps, B, n = 100, 64, 4
    if kind == 'vectorspace':
        z, Vw, Ve, dw, de = 5, 300, 25, 32, 32
        p = U.make_vs_problem(71, B * 4, n, z, Vw, Ve, dw, de)
        eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        ora = O.VectorSpaceOracle(B, n, z, p['Rw'], p['Re'], p['W'], p['b'], 0.01)
        for s in range(steps):
            j = s % 4
            sl = slice(j * B, (j + 1) * B)
            neg = p['rng'].randint(0, Ve, (B, z)).astype(np.int64)
            ref = ora.train_step(p['X'][sl], p['y'][sl], p['w'][sl], neg)
            loss = eng.train_batch(j, neg)
            assert abs(loss - ref) <= 2e-5 * abs(ref), (s, loss, ref)
        pairs = [(C.T_RW, ora.R_w), (C.T_RE, ora.R_e), (C.T_W, ora.W), (C.T_B, ora.b)]
    else:
        Vw, Ve, d = 300, 40, 24
        p = U.make_ll_problem(72, B * 4, n, Vw, Ve, d, 'csr')
        eng = U.ll_engine(p, B, n, 0.01, keep_grads=0)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], csr=p['y'], w=p['w'])
        ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], 0.01)
        for s in range(steps):
            j = s % 4
            sl = slice(j * B, (j + 1) * B)
            ref = ora.train_step(p['X'][sl], p['ydense'][sl], p['w'][sl])
            loss = eng.train_batch(j)
            assert abs(loss - ref) <= 2e-5 * abs(ref), (s, loss, ref)
        pairs = [(C.T_RW, ora.R_w), (C.T_W, ora.W), (C.T_B, ora.b)]
    for which, ref in pairs:
        assert U.rel_err(eng.get_tensor(which), np.asarray(ref).ravel()) < 1e-4, which
    eng.close()


@pytest.mark.gpu
def test_loglinear_dg_rows_stored_through_the_row_map(hip_lib):
    """Without keep_grads and with a dG = dZ.W^T that the 64x64-tile GEMM takes (U x d >= 16384 elements, fewer than
    512 big tiles), the GEMM's epilogue stores row u of dG as row uwords[u] of the word-table gradient -- no scatter
    pass.  Losses of five steps and the parameters after them against the oracle (the word-table rows of every
    distinct word of a batch must have received exactly their gradient row)."""
    B, n, Vw, Ve, d, steps = 256, 6, 2000, 60, 32, 5
    p = U.make_ll_problem(81, B * 2, n, Vw, Ve, d, 'int')
    assert len(np.unique(p['X'][:B])) * d >= 4 * 64 * 64
    eng = U.ll_engine(p, B, n, 0.01, keep_grads=0)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], 0.01)
    for s in range(steps):
        j = s % 2
        sl = slice(j * B, (j + 1) * B)
        ref = ora.train_step(p['X'][sl], p['ydense'][sl], p['w'][sl])
        loss = eng.train_batch(j)
        assert abs(loss - ref) <= LOSS_TOL * abs(ref), (s, loss, ref)
    for which, ref in ((C.T_RW, ora.R_w), (C.T_W, ora.W), (C.T_B, ora.b)):
        assert U.rel_err(eng.get_tensor(which), np.asarray(ref).ravel()) < PARAM_TOL, which
    # Adadelta's accumulators of the word table: a row that took a wrong gradient row shows here first
    assert U.rel_err(eng.get_tensor(C.T_STATE0_RW), ora.opt.accu[0].ravel()) < PARAM_TOL
    eng.close()


def test_scratch_buffers_are_not_readable_without_keep_grads(hip_lib):
    B, n, z = 16, 2, 2
    p = U.make_vs_problem(81, B, n, z, 30, 6, 8, 8)
    eng = U.vs_engine(p, B, n, z, 0.0, keep_grads=0)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    eng.train_batch(0)
    for which in (C.T_GRAD_RW, C.T_GRAD_W, C.T_ACT_H):
        with pytest.raises(C.SertError):
            eng.get_tensor(which)
    assert np.all(np.isfinite(eng.get_tensor(C.T_RW)))
    eng.close()


@pytest.mark.parametrize('dims', [
    dict(B=16, n=4, Vw=120, Ve=200, d=16),        # fused LDS path
    dict(B=4, n=4, Vw=120, Ve=9004, d=12),        # streaming path
])
@pytest.mark.parametrize('keep', [1, 0])
def test_loglinear_with_saturated_probabilities(hip_lib, dims, keep):
    """Large logits: many token probabilities fall below eps = 1e-7 (and one sits above
    1 - eps), so the clip masks of models.py:200 are exercised in the loss AND in the
    backward (mask_ke in dZ = mask dJ - P r).  keep=1: the per-token path with gradient
    checks; keep=0: the distinct-word path, parameters after 3 steps."""
    B, n = dims['B'], dims['n']
    p = U.make_ll_problem(91, B * 3, n, dims['Vw'], dims['Ve'], dims['d'], 'int')
    p['W'] = (p['W'] * 40.0).astype(np.float32)          # logits of magnitude ~10-30
    p['b'] = (p['b'] * 30.0).astype(np.float32)
    eng = U.ll_engine(p, B, n, 0.01, keep_grads=keep)
    eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
    ora = O.LogLinearOracle(B, n, p['Rw'], p['W'], p['b'], 0.01)
    _, Pref = ora.token_distributions(p['X'][:B])
    assert (Pref < 1e-7).mean() > 0.05                    # the masks really are active
    for s in range(3):
        sl = slice(s * B, (s + 1) * B)
        loss_ref, grads_ref, _ = ora.loss_and_grads(p['X'][sl], p['ydense'][sl], p['w'][sl])
        ora.opt.update(ora.params(), grads_ref)
        loss = eng.train_batch(s)
        assert abs(loss - loss_ref) <= 2e-5 * abs(loss_ref), (s, loss, loss_ref)
        if keep:
            dRw, dW, db = grads_ref
            assert U.rel_err(eng.get_tensor(C.T_GRAD_W), dW.ravel()) < GRAD_TOL
            assert U.rel_err(eng.get_tensor(C.T_GRAD_B), db.ravel()) < GRAD_TOL
            assert U.rel_err(eng.get_tensor(C.T_GRAD_RW), dRw.ravel()) < GRAD_TOL
    assert U.rel_err(eng.get_tensor(C.T_W), ora.W.ravel()) < PARAM_TOL
    assert U.rel_err(eng.get_tensor(C.T_RW), ora.R_w.ravel()) < PARAM_TOL
    eng.close()


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_run_ahead_randomised_differential(hip_lib, kind):
    """Random interleavings of training steps (with right, wrong or no announcements),
    evaluations, parameter writes/reads and multi-step calls: an engine that is told what comes
    next must return exactly what an engine that is never told anything returns."""
    B, n, z, Vw, Ve, d = 32, 3, 4, 150, 10, 16
    nb = 6
    if kind == 'vectorspace':
        p = U.make_vs_problem(97, B * nb, n, z, Vw, Ve, d, d)
        mk = lambda: U.vs_engine(p, B, n, z, 0.02, keep_grads=0, seed=5)
    else:
        p = U.make_ll_problem(97, B * nb, n, Vw, Ve, d, 'int')
        mk = lambda: U.ll_engine(p, B, n, 0.02, keep_grads=0)
    rng = np.random.RandomState(123)
    ops = []
    for _ in range(60):
        r = rng.rand()
        if r < 0.6:
            ops.append(('train', int(rng.randint(nb)), rng.choice(['right', 'wrong', 'none'])))
        elif r < 0.72:
            ops.append(('eval', int(rng.randint(nb))))
        elif r < 0.80:
            ops.append(('set_b',))
        elif r < 0.88:
            ops.append(('get',))
        else:
            ops.append(('many', [int(v) for v in rng.randint(nb, size=3)]))
    # "right" announces the batch of the next train op (if the next op is one)
    results = []
    for hinted in (False, True):
        eng = mk()
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        out = []
        for i, op in enumerate(ops):
            if op[0] == 'train':
                if hinted and op[2] != 'none':
                    nxt = ops[i + 1] if i + 1 < len(ops) else None
                    if op[2] == 'right' and nxt is not None and nxt[0] == 'train':
                        eng.hint_next_batch(nxt[1])
                    elif op[2] == 'right' and nxt is not None and nxt[0] == 'many':
                        eng.hint_next_batch(nxt[1][0])
                    else:
                        eng.hint_next_batch((op[1] + 1 + i) % nb)
                out.append(float(eng.train_batch(op[1])))
            elif op[0] == 'eval':
                out.append(float(eng.eval_batch(C.SPLIT_TRAIN, op[1])))
            elif op[0] == 'set_b':
                b = eng.get_tensor(C.T_B)
                eng.set_tensor(C.T_B, (b * 0.5).astype(np.float32))
            elif op[0] == 'get':
                out.append(float(np.abs(eng.get_tensor(C.T_RW)).sum()))
            else:
                out.extend(float(v) for v in eng.train_batches(op[1]))
        out.append(eng.get_tensor(C.T_RW).copy())
        out.append(eng.get_tensor(C.T_W).copy())
        eng.close()
        results.append(out)
    a, b = results
    assert a[:-2] == b[:-2]
    assert np.array_equal(a[-2], b[-2]) and np.array_equal(a[-1], b[-1])


@pytest.mark.parametrize('kind', ['vectorspace', 'loglinear'])
def test_eval_batches_equals_eval_batch_loop(hip_lib, kind):
    """sert_eval_batches (one host synchronisation per chunk -- what train_error() and
    validation_error() use) returns exactly the per-batch values of sert_eval_batch, in order,
    and leaves the sampler / model state identical."""
    B, n, nb = 64, 4, 7
    if kind == 'vectorspace':
        p = U.make_vs_problem(41, B * nb, n, 5, 300, 40, 16, 16)
        mk = lambda: U.vs_engine(p, B, n, 5, 0.01, keep_grads=0)
    else:
        p = U.make_ll_problem(41, B * nb, n, 300, 40, 16)
        mk = lambda: U.ll_engine(p, B, n, 0.01, keep_grads=0)
    out = []
    for mode in ('loop', 'batched'):
        eng = mk()
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        eng.upload_dataset(C.SPLIT_VALIDATE, p['X'][:B * 3], y_int=p['y'][:B * 3])
        eng.train_batch(0)
        order = [3, 0, 6, 1, 1, 5]
        if mode == 'loop':
            a = np.array([eng.eval_batch(C.SPLIT_TRAIN, i) for i in order], np.float32)
            b = np.array([eng.eval_batch(C.SPLIT_VALIDATE, i) for i in (2, 0)], np.float32)
        else:
            a = eng.eval_batches(C.SPLIT_TRAIN, order)
            b = eng.eval_batches(C.SPLIT_VALIDATE, [2, 0])
        c = eng.train_batch(2)          # training continues identically afterwards
        out.append((a, b, c))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]
    assert np.all(np.isfinite(out[0][0]))


SCHEDULE_WORKER = r'''
import sys, json, zlib
sys.path.insert(0, %(root)r)
import numpy as np
from tests import util as U
from sert_amd import _capi as C
B, n, z, Vw, Ve, d = 8192, 10, 10, 20000, 1000, 128
p = U.make_vs_problem(3, 2 * B, n, z, Vw, Ve, d, d, zipf=True)
eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0, seed=11)
eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
out = {'losses': [float(eng.train_batch(s %% 2)) for s in range(6)]}
for name, which in (('Rw', C.T_RW), ('Re', C.T_RE), ('W', C.T_W), ('b', C.T_B),
                    ('m_Re', C.T_STATE0_RE), ('v_W', C.T_STATE1_W)):
    out['crc_' + name] = zlib.crc32(eng.get_tensor(which).tobytes())
eng.close()
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.gpu
def test_schedule_variants_do_not_change_a_bit(hip_lib):
    """The stream schedule is only a schedule: one stream, the side-heavy schedule (and, against a variants
    build: the round-1 three-event schedule, plain event records instead of kernel stop events, the entity group
    sum in a launch of its own, the fork behind the NCE kernel, the two upper levels of the word-gradient tree as
    two launches, the three-launch tail) -- six steps (device-drawn negatives, pre-drawn on the side stream where there is one)
    end in bit-identical parameters, optimiser state and losses, each in a fresh process (the
    knobs are read once per process)."""
    import json
    import os
    import subprocess
    import sys
    code = SCHEDULE_WORKER % dict(root=U.ROOT)
    variants = ({}, {'SERT_STREAMS': '1'}, {'SERT_ROCTX': '1'},     # (a roctx range around every kernel group)
                {'SERT_SIDE_HEAVY': '2'})   # (entity chain, dW and the small-tensor update on the side stream)
    if VARIANTS_BUILD:      # knobs only a -DSERT_VARIANTS library reads (common.h: variant_knob)
        variants += ({'SERT_FORK_LATE': '0'}, {'SERT_EXT_EVENTS': '0'}, {'SERT_EGRAD_GROUP_SUM': '1'}, {'SERT_FORK_AT': 'nce'}, {'SERT_FORK_AT': 'nce_dw'},
                     {'SERT_DW_FIRST': '0'},      # (dW / db on the main stream instead of first on the side stream)
                     {'SERT_SEG_NO_FUSED_UPPER': '1'}, {'SERT_NO_TAIL': '1'},
                     {'SERT_NO_EARLY_BUCKET': '1'}, {'SERT_NO_EARLY_SORT': '1'})    # (round 6: the entity keys' partition behind the fork again instead of beside the forward)
    outs = []
    for extra in variants:
        r = subprocess.run([sys.executable, '-c', code], check=True, env=dict(os.environ, **extra),
                           cwd=U.ROOT, stdout=subprocess.PIPE, timeout=600)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[len('RESULT '):]))
    for extra, o in zip(variants[1:], outs[1:]):
        assert o == outs[0], 'schedule variant %r changed the results' % (extra,)
    assert all(np.isfinite(outs[0]['losses']))


LL_SCHEDULE_WORKER = r'''
import sys, json, zlib
sys.path.insert(0, %(root)r)
import numpy as np
from tests import util as U
from sert_amd import _capi as C
B, n, Vw, Ve, d = %(dims)s
p = U.make_ll_problem(5, 3 * B, n, Vw, Ve, d, 'int')
eng = U.ll_engine(p, B, n, 0.01, keep_grads=0)
eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
losses = []
for s in range(6):
    eng.hint_next_batch((s + 1) %% 3 if s < 5 else -1)     # (the next batch's forward + backward run ahead)
    losses.append(float(eng.train_batch(s %% 3)))
out = {'losses': losses}
for name, which in (('Rw', C.T_RW), ('W', C.T_W), ('b', C.T_B), ('a_W', C.T_STATE0_W), ('d_b', C.T_STATE1_B)):
    out['crc_' + name] = zlib.crc32(eng.get_tensor(which).tobytes())
eng.close()
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.gpu
@pytest.mark.parametrize('dims', ['4096, 8, 5000, 2000, 128', '2048, 8, 20000, 40000, 128'])
def test_loglinear_side_stream_does_not_change_a_bit(hip_lib, dims):
    """Loglinear with a dW GEMM worth forking for (2 x 5000 x 128 x 2000 flop): dW, its combine and the W, b
    update on the side stream, with the next batch running ahead, against the whole step on one stream
    (SERT_LL_DW_SIDE=0, SERT_STREAMS=1): bit-identical losses, parameters and Adadelta state.  Second case: a W of
    128 x 40 000 = 5.1 M elements -- a "big tensor" with a streaming launch of its own; updated on the main stream it read
    dW's gradient while the side stream was still writing it (round 4: found at C4's sizes, fixed)."""
    import json
    import os
    import subprocess
    import sys
    code = LL_SCHEDULE_WORKER % dict(root=U.ROOT, dims=dims)
    variants = ({}, {'SERT_LL_DW_SIDE': '0'}, {'SERT_STREAMS': '1'}, {})     # (the default twice: run-to-run too)
    outs = []
    for extra in variants:
        r = subprocess.run([sys.executable, '-c', code], check=True, env=dict(os.environ, **extra),
                           cwd=U.ROOT, stdout=subprocess.PIPE, timeout=600)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[len('RESULT '):]))
    for extra, o in zip(variants[1:], outs[1:]):
        assert o == outs[0], 'schedule variant %r changed the results' % (extra,)
    assert all(np.isfinite(outs[0]['losses']))


@pytest.mark.parametrize('dims', [
    dict(B=4096, n=4, z=6, Vw=3000, Ve=5000, dw=32, de=64),       # sorted entity chain (V_e > 2048): counting sort, chunked reduce, fix-up
    dict(B=2048, n=3, z=10, Vw=2000, Ve=300, dw=64, de=128),      # sort-free chain (per-group partial tables)
    dict(B=700, n=2, z=3, Vw=500, Ve=40000, dw=16, de=16),        # far more entities than pairs: most rows of dR_e come from the fix-up's zero fill
])
@pytest.mark.parametrize('ranges', [False, True])
def test_steps_read_nothing_stale_from_the_gradient_scratch(hip_lib, monkeypatch, dims, ranges):
    """A vectorspace step whose negatives were drawn at the end of the previous step launches NO prologue: nothing zeroes
    the flat gradient buffer or the per-entity run bounds (sert_hip.hip: step_forward_backward, have_neg).  That is right only
    while every value the step reads from there was written by the step's own kernels -- every row of dR_e by the entity
    chain, the run bounds by the sort's first histogram pass, g_W / g_b by the split-K combine (advisor, round 5).  Checked
    by force: quiet NaNs into the whole buffer and wrong bounds behind it between the steps (sert_debug_poison_scratch)
    must change NOTHING -- losses, both tables, bit for bit."""
    if ranges:
        if not VARIANTS_BUILD:
            pytest.skip('SERT_EGRAD_RANGES is read only by a library built with -DSERT_VARIANTS')
        monkeypatch.setenv('SERT_EGRAD_RANGES', '1')
    B, n, z, Vw, Ve, dw, de = (dims[k] for k in ('B', 'n', 'z', 'Vw', 'Ve', 'dw', 'de'))
    p = U.make_vs_problem(29, 3 * B, n, z, Vw, Ve, dw, de, zipf=True)
    outs = []
    for poison in (False, True):
        eng = U.vs_engine(p, B, n, z, 0.01, keep_grads=0, seed=5)
        eng.upload_dataset(C.SPLIT_TRAIN, p['X'], y_int=p['y'], w=p['w'])
        losses = []
        for s in range(6):
            losses.append(eng.train_batch(s % 3))          # (device-drawn negatives, no hint: no run-ahead in flight)
            if poison:
                eng.poison_scratch()
        outs.append((losses, eng.get_tensor(C.T_RW).copy(), eng.get_tensor(C.T_RE).copy(), eng.get_tensor(C.T_W).copy(),
                     eng.get_tensor(C.T_B).copy()))
        eng.close()
    assert all(np.isfinite(outs[1][0])), outs[1][0]
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)
