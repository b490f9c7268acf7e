import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'jitter_expected: the test drives the reference\'s 1e-7 jitter branch on purpose (partial_terms.py:452-456)')


@pytest.fixture(autouse=True)
def _no_silent_jitter(request):
    """A GPU test that takes the reference's jitter branch without saying so fails (round 6).  The branch is silent by design -- a failed factorisation is repeated with
    1e-7 on the diagonal and the evaluation goes on -- and that is how a WRONG factorisation hides: round 5's one unreproduced failure was a spurious failure flag from a
    race in the blocked Cholesky, then a correct evaluation of the jittered problem, 9.97e-5 off in grad_Z and inside the tolerance everywhere else
    (profiles/r06_first_evaluation_race.txt).  Every retry goes through gp_global_step_jitter(h, mask != 0), whatever the host surface: counted here, in-process."""
    if 'gpu' not in request.keywords:
        yield
        return
    from gparml_amd import _lib
    lib = _lib.load()
    orig = lib.gp_global_step_jitter
    masks = []

    def counted(h, mask):
        if mask:
            masks.append(int(mask))
        return orig(h, mask)
    lib.gp_global_step_jitter = counted
    try:
        yield
    finally:
        lib.gp_global_step_jitter = orig
    if masks and request.node.get_closest_marker('jitter_expected') is None:
        pytest.fail('the 1e-7 jitter retry was taken %d time(s) (masks %s) in a test that is not marked jitter_expected: a factorisation failed where none should'
                    % (len(masks), sorted(set(masks))))


# Order of the GPU suite under `-x` (round-4 review, "What's weak" 2): the golden / oracle parity files first -- they carry the
# parity claim and must not sit behind a multi-process launch or a 49 GB allocation --, then the size-independent properties at
# full size, the multi-process tests (2 and 8 torch processes on one device) last.  Files not listed keep their alphabetical place in
# tier 0 behind the listed ones; the order inside a file is untouched.
_GPU_ORDER = [
    # tier 0: golden fixtures of the reference and the oracle, method by method, then the extended-precision truths
    'test_gpu_parity', 'test_gpu_partial_terms', 'test_gpu_pipeline', 'test_gpu_predict', 'test_gpu_phase2_general',
    'test_gpu_hp_truth', 'test_hp_truth_large', 'test_gpu_global_step', 'test_gpu_linalg',
    'test_gpu_c_consumer', 'test_gpu_first_evaluation', 'test_gpu_resident_scg', 'test_gpu_resident_gd', 'test_gpu_dropout', 'test_gpu_tile_phase2', 'test_gpu_p1_i8', 'test_gpu_fuzz_shapes',
]
_GPU_LATE = [
    # tier 1: properties at the configurations' full sizes (seconds of device time, gigabytes of host data)
    'test_gpu_fullsize', 'test_gpu_index_range', 'test_gpu_config4_scg', 'test_gpu_config4_fullsize',
    # tier 2: several torch processes
    'test_gpu_bench_ranks',
    # last: absolute kernel durations (a contended or throttled box must not stop the parity files under -x)
    'test_gpu_perf_guard',
]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _GPU_ORDER:
            return (0, _GPU_ORDER.index(name))
        if name in _GPU_LATE:
            return (2, _GPU_LATE.index(name))
        return (1, 0)
    items.sort(key=key)            # stable: collection order is kept inside a file and among unlisted files
    # GPARML_TEST_ORDER=reversed | shuffle:<seed>: the FILES in another order (order inside a file kept) -- round 6's hunt for a failure that was seen
    # once in a full-suite process and never in isolation (tools/suite_orders.sh)
    mode = os.environ.get('GPARML_TEST_ORDER', '')
    if mode:
        files = []
        for it in items:
            f = str(it.fspath)
            if f not in files:
                files.append(f)
        if mode == 'reversed':
            files.reverse()
        elif mode.startswith('shuffle:'):
            import random
            random.Random(int(mode.split(':')[1])).shuffle(files)
        items.sort(key=lambda it: files.index(str(it.fspath)))


def golden_names():
    return sorted(f[3:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith('pt_') and f.endswith('.npz'))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, 'pt_%s.npz' % name))
    inp = {k[3:]: z[k] for k in z.files if k.startswith('in_')}
    out = {k[4:]: z[k] for k in z.files if k.startswith('out_')}
    for k in ('N', 'D', 'M', 'Q'):
        inp[k] = int(inp[k])
    for k in ('sf2', 'beta'):
        inp[k] = float(inp[k])
    return inp, out


@pytest.fixture(params=golden_names())
def golden(request):
    inp, out = load_golden(request.param)
    return request.param, inp, out


def assert_close(a, b, rtol, atol=0.0, what=''):
    if hasattr(b, 'check') and hasattr(b, 'rows'):       # tests/oracle_cache.py Sampled: a per-point oracle output kept as exact rows + two projections of every row
        b.check(a, rtol, what)
        import oracle_cache
        r = oracle_cache.LAST_RATIOS.get(what)
        if r:      # what the device actually needs of the two bounds (pytest -s / a failing test shows it): the bounds are set from these
            print('[oracle cache] %s: sampled rows at %.3g of the bound, worst row projection at %.3g of its bound (rtol %.1e)' % (what, r[0], r[1], rtol))
        return
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 0.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    assert err <= rtol * scale + atol, '%s: max abs err %.3e, scale %.3e (rel %.3e > %.1e)' % (
        what, err, scale, err / max(scale, 1e-300), rtol)
