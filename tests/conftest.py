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
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 0.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    assert err <= rtol * scale + atol, '%s: max abs err %.3e, scale %.3e (rel %.3e > %.1e)' % (
        what, err, scale, err / max(scale, 1e-300), rtol)
