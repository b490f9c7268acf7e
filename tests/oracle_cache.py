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
doubles; Psi2 and Psi1^T Y at M = 1024) are stored as ``Sampled``: evenly spaced rows plus the first and last rows of every 128-row tile
boundary region exactly (at most 1024 rows / 32768 elements), and for EVERY row two signed projections onto fixed weight vectors over the columns
(2 doubles per row: a wrong element anywhere -- a tail tile, a shard-boundary row -- moves its row's projections; round 5 kept four weighted sums
over all rows instead, which let a localised error of ~1e3 x rtol through).  conftest.assert_close understands ``Sampled``.
Nested dictionaries (the oracle's ``stats``) are stored under ``outer.inner`` names.

Audit (round 6): ``GPARML_AUDIT_ORACLE_CACHE=1`` makes get() run the oracle live AND compare it with the committed file at 1e-9 (then done()
skips the device part, as in writing mode); ``GPARML_AUDIT_KEYS=a,b`` restricts it to those files.  tests/test_oracle_cache.py runs this
over three files in the CPU suite by default and over every file with GPARML_AUDIT_ORACLE_CACHE_ALL=1 (once per round: profiles/).

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


def _col_weights(cols):
    """Two fixed weight vectors over the columns (bounded by 1, never both small at the same column)."""
    j = np.arange(cols, dtype=np.float64)
    return np.stack([np.cos(0.9 * j + 0.1), np.sin(0.9 * j + 0.1)])


LAST_RATIOS = {}     # what -> (worst sampled-row error, worst per-row projection error) / (rtol * scale): printed by the GPU tests, sets the bounds from data


class Sampled(object):
    """A per-point array (n rows) kept as selected rows exactly + two signed projections of EVERY row + its max-norm."""

    def __init__(self, n, rows, values, rowp, scale):
        self.n, self.rows, self.values, self.rowp, self.scale = int(n), rows, values, rowp, float(scale)

    @classmethod
    def of(cls, a):
        a = np.asarray(a, dtype=np.float64)
        n = a.shape[0]
        a2 = a.reshape(n, -1)
        cols = a2.shape[1]
        budget = min(n, MAX_ROWS, max(16, MAX_ELEMS // cols))
        # rows next to the 128-row tile boundaries at both ends of the array (tail tiles, shard boundaries) first, evenly spaced rows for the rest
        edge = [r for r in (0, 1, 127, 128, 129, n - 130, n - 129, n - 128, n - 2, n - 1, (n // 128) * 128 - 1, (n // 128) * 128) if 0 <= r < n]
        rows = np.unique(np.concatenate([np.array(edge, dtype=np.int64), np.linspace(0, n - 1, max(2, budget - len(edge))).astype(np.int64)]))
        return cls(n, rows, a[rows].copy(), a2.dot(_col_weights(cols).T), np.max(np.abs(a)))

    def check(self, a, rtol, what=''):
        """The stored rows to rtol * max-norm per element; both projections of EVERY row to rtol * max-norm * ||w||_1 (what element errors of at most
        rtol * max-norm can add up to in a row: a necessary condition for every row, and an element that is off by more than about `cols` times
        the tolerance anywhere in the array violates it)."""
        a = np.asarray(a, dtype=np.float64)
        assert a.shape[0] == self.n and a[self.rows].shape == self.values.shape, '%s: shape %s vs %d rows' % (what, a.shape, self.n)
        err = np.max(np.abs(a[self.rows] - self.values))
        assert err <= rtol * self.scale, '%s: sampled rows: max abs err %.3e, scale %.3e (rel %.3e > %.1e)' % (what, err, self.scale, err / self.scale, rtol)
        a2 = a.reshape(self.n, -1)
        w = _col_weights(a2.shape[1])
        perr = np.abs(a2.dot(w.T) - self.rowp)
        bound = rtol * self.scale * np.abs(w).sum(axis=1)
        worst = int(np.argmax(np.max(perr / bound, axis=1)))
        LAST_RATIOS[what] = (err / (rtol * self.scale), float(np.max(perr[worst] / bound)))
        assert np.all(perr <= bound), '%s: row %d of %d: projection error %s > %s (a row that is not among the exactly stored ones is off)' % (
            what, worst, self.n, perr[worst], bound)
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
    auditing = bool(os.environ.get('GPARML_AUDIT_ORACLE_CACHE')) and not writing
    audit_keys = [k for k in os.environ.get('GPARML_AUDIT_KEYS', '').split(',') if k]
    if os.path.exists(path) and not writing:
        z = np.load(path)
        if z['input_checksums'].shape == cs.shape and np.allclose(cs, z['input_checksums'], rtol=1e-11, atol=1e-9):
            out = {}
            for k in z.files:
                if k.startswith('out_'):
                    out[k[4:]] = z[k] if z[k].ndim else float(z[k])
                elif k.startswith('smp_rows_'):
                    name = k[9:]
                    out[name] = Sampled(int(z['smp_n_' + name]), z[k], z['smp_values_' + name], z['smp_rowp_' + name], float(z['smp_scale_' + name]))
            stored = _unflatten(out)
            if not auditing:
                return stored
            if audit_keys and key not in audit_keys:
                pytest.skip('oracle cache audit: %s is not among GPARML_AUDIT_KEYS' % key)
            import time
            t = time.time()
            _audit(key, _flatten(stored), _flatten(compute()))
            print('ORACLE_CACHE_AUDIT %s ok: live oracle (%.0f s) agrees with the committed file at 1e-9' % (key, time.time() - t), flush=True)
            return stored
        assert not auditing, 'oracle cache audit: %s does not describe the inputs the test builds (checksums differ)' % path
    assert not (auditing and not writing), 'oracle cache audit: %s is missing' % path
    ref = compute()
    if writing:
        rec = {'input_checksums': cs}
        for k, v in _flatten(ref).items():
            if k.startswith('gstep.'):         # the oracle's intermediate M x M matrices: no test that uses the cache compares them
                continue
            if isinstance(v, np.ndarray) and v.ndim == 2 and v.size > MAX_ELEMS:
                s = Sampled.of(v)
                rec.update({'smp_rows_' + k: s.rows, 'smp_values_' + k: s.values, 'smp_rowp_' + k: s.rowp, 'smp_scale_' + k: np.float64(s.scale),
                            'smp_n_' + k: np.int64(s.n)})
            elif isinstance(v, (int, float, np.floating, np.ndarray)):
                rec['out_' + k] = np.asarray(v, dtype=np.float64)
        os.makedirs(CACHE_DIR, exist_ok=True)
        np.savez_compressed(path, **rec)
        print('oracle cache written: %s (%d bytes)' % (path, os.path.getsize(path)))
    return ref


def _audit(key, stored, live):
    """Every stored output of ``key`` against the oracle run now: scalars and small arrays at 1e-9 of their max-norm, Sampled arrays through check()."""
    from conftest import assert_close
    for k, v in stored.items():
        assert k in live, 'oracle cache audit %s: the live oracle has no output %s' % (key, k)
        assert_close(live[k], v, 1e-9, atol=1e-300, what='%s: %s (live oracle vs committed file)' % (key, k))


def done():
    """Called by a test after its last get() and before it touches the device: in writing and in audit mode the test ends here."""
    if os.environ.get('GPARML_WRITE_ORACLE_CACHE'):
        pytest.skip('oracle cache written (GPARML_WRITE_ORACLE_CACHE): the device part of the test is not run')
    if os.environ.get('GPARML_AUDIT_ORACLE_CACHE'):
        pytest.skip('oracle cache audited (GPARML_AUDIT_ORACLE_CACHE): the device part of the test is not run')
