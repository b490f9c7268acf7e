"""Committed outputs of the CPU oracle for the GPU tests whose oracle evaluation is minutes of host time.

Why: the GPU suite runs under a 1200 s limit on boxes whose hosts differ by a factor of four; 260 of its 420 s (on a fast box) were the
oracle -- numpy on the host -- recomputing the same seeded cases on every run (tests/golden/... holds the reference's own vectors; these
are the ORACLE's vectors for cases the reference cannot run: it stores an (N, M, M) tensor, partial_terms.py:45).

How: a test builds its seeded inputs as before and asks ``get(key, inputs, compute)`` for the oracle's outputs.
  * tests/golden/oracle_cache/<key>.npz exists and the checksums of the regenerated inputs match the stored ones (1e-11: sin() may differ in
    the last bit between hosts) -> the stored outputs are returned;
  * otherwise ``compute()`` runs the oracle live, exactly as before the cache existed (a changed case silently falls back to the live oracle;
    nothing is ever compared with numbers that belong to other inputs).
Small outputs (bound, hyper-parameter gradients) are stored whole.  Outputs of more than 32768 elements (grad_X_mu, grad_X_S: up to 1e5 x 10
doubles; Psi2 and Psi1^T Y at M = 1024) are stored as ``Sampled``: evenly spaced rows exactly (at most 1024 rows / 32768 elements), plus four
fixed weighted row sums of the WHOLE array, so that every row still enters the comparison (conftest.assert_close understands ``Sampled``).
Nested dictionaries (the oracle's ``stats``) are stored under ``outer.inner`` names.

Regenerate (CPU only, no GPU needed; minutes on 8 cores):
    GPARML_WRITE_ORACLE_CACHE=1 python -m pytest tests -m gpu -q -k "<the tests that call get()>"
Every test calls get() and then done() BEFORE it touches the device; with the variable set, get() computes and writes the file and done()
skips the rest of the test.
"""
import os

import numpy as np
import pytest

CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_cache')
MAX_ROWS = 1024
MAX_ELEMS = 32768


def _weights(n):
    """Four fixed weight vectors over n rows (bounded by 1, different frequencies): every row enters every weighted sum."""
    i = np.arange(n, dtype=np.float64)
    return np.stack([np.cos(0.7 * i + 0.3), np.sin(1.3 * i), np.cos(0.011 * i), np.ones(n)])


class Sampled(object):
    """A per-point array (n rows) kept as evenly spaced rows + weighted sums over all rows + its max-norm."""

    def __init__(self, n, rows, values, proj, scale):
        self.n, self.rows, self.values, self.proj, self.scale = int(n), rows, values, proj, float(scale)

    @classmethod
    def of(cls, a):
        a = np.asarray(a, dtype=np.float64)
        n = a.shape[0]
        cols = max(1, a.size // max(n, 1))
        rows = np.unique(np.linspace(0, n - 1, min(n, MAX_ROWS, max(16, MAX_ELEMS // cols))).astype(np.int64))
        return cls(n, rows, a[rows].copy(), _weights(n).dot(a.reshape(n, -1)), np.max(np.abs(a)))

    def check(self, a, rtol, what=''):
        """The stored rows to rtol * max-norm each; the weighted sums over ALL rows to rtol * max-norm * 4 sqrt(n) (errors of
        independent sign add up like sqrt(n); a single wrong row of relative size 4 rtol sqrt(n) / 1 is seen, anything systematic much earlier)."""
        a = np.asarray(a, dtype=np.float64)
        assert a.shape[0] == self.n and a[self.rows].shape == self.values.shape, '%s: shape %s vs %d rows' % (what, a.shape, self.n)
        err = np.max(np.abs(a[self.rows] - self.values))
        assert err <= rtol * self.scale, '%s: sampled rows: max abs err %.3e, scale %.3e (rel %.3e > %.1e)' % (what, err, self.scale, err / self.scale, rtol)
        perr = np.max(np.abs(_weights(self.n).dot(a.reshape(self.n, -1)) - self.proj))
        bound = rtol * self.scale * 4.0 * np.sqrt(self.n)
        assert perr <= bound, '%s: weighted row sums over all %d rows: err %.3e > %.3e' % (what, self.n, perr, bound)
        m = np.max(np.abs(a))
        assert abs(m - self.scale) <= 1e-3 * self.scale, '%s: max-norm %.6e vs %.6e' % (what, m, self.scale)


def checksums(inputs):
    out = []
    for k in sorted(inputs):
        a = np.atleast_1d(np.asarray(inputs[k], dtype=np.float64))
        flat = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
        w = np.cos(np.arange(flat.shape[0], dtype=np.float64))
        out += [flat.sum(), np.abs(flat).sum(), w.dot(flat).sum(), float(a.size)]
    return np.array(out)


def _flatten(d, prefix=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + '.'))
        else:
            out[prefix + k] = v
    return out


def _unflatten(flat):
    out = {}
    for k, v in flat.items():
        parts = k.split('.')
        t = out
        for p in parts[:-1]:
            t = t.setdefault(p, {})
        t[parts[-1]] = v
    return out


def get(key, inputs, compute):
    """Oracle outputs for the seeded case ``key``: from the committed file when it describes exactly these inputs, live otherwise."""
    path = os.path.join(CACHE_DIR, key + '.npz')
    cs = checksums(inputs)
    writing = bool(os.environ.get('GPARML_WRITE_ORACLE_CACHE'))
    if os.path.exists(path) and not writing:
        z = np.load(path)
        if z['input_checksums'].shape == cs.shape and np.allclose(cs, z['input_checksums'], rtol=1e-11, atol=1e-9):
            out = {}
            for k in z.files:
                if k.startswith('out_'):
                    out[k[4:]] = z[k] if z[k].ndim else float(z[k])
                elif k.startswith('smp_rows_'):
                    name = k[9:]
                    out[name] = Sampled(int(z['smp_n_' + name]), z[k], z['smp_values_' + name], z['smp_proj_' + name], float(z['smp_scale_' + name]))
            return _unflatten(out)
    ref = compute()
    if writing:
        rec = {'input_checksums': cs}
        for k, v in _flatten(ref).items():
            if k.startswith('gstep.'):         # the oracle's intermediate M x M matrices: no test that uses the cache compares them
                continue
            if isinstance(v, np.ndarray) and v.ndim == 2 and v.size > MAX_ELEMS:
                s = Sampled.of(v)
                rec.update({'smp_rows_' + k: s.rows, 'smp_values_' + k: s.values, 'smp_proj_' + k: s.proj, 'smp_scale_' + k: np.float64(s.scale),
                            'smp_n_' + k: np.int64(s.n)})
            elif isinstance(v, (int, float, np.floating, np.ndarray)):
                rec['out_' + k] = np.asarray(v, dtype=np.float64)
        os.makedirs(CACHE_DIR, exist_ok=True)
        np.savez_compressed(path, **rec)
        print('oracle cache written: %s (%d bytes)' % (path, os.path.getsize(path)))
    return ref


def done():
    """Called by a test after its last get() and before it touches the device: in writing mode the test ends here."""
    if os.environ.get('GPARML_WRITE_ORACLE_CACHE'):
        pytest.skip('oracle cache written (GPARML_WRITE_ORACLE_CACHE): the device part of the test is not run')
