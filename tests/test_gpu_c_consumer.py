"""The C ABI used from plain C (examples/c_consumer.c): no Python, no torch in the consumer process.  The test compiles the example against
include/gparml_hip.h, runs it on a seeded shard and compares what it prints with the Python engine (bit-identical: both drive the same library the same
way) and with the oracle (BASELINE.json's tolerances).  Reference counterpart of the sequence: /root/reference/partial_terms.py:38-52, 207-360, 436-473."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import assert_close

F_RTOL, G_RTOL = 1e-6, 1e-5          # BASELINE.json: bound / gradients

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer_matches_the_python_engine_and_the_oracle(tmp_path):
    if shutil.which('gcc') is None:
        pytest.skip('no gcc on this host')
    from gparml_amd.engine import ShardEngine
    from oracle import factorised as Fz
    N, D, M, Q = 1500, 6, 200, 11          # a shape of tests/test_gpu_parity.py's fixed-embedding list
    d = Fz.synthetic_shard(N, D, M, Q, regime='A', seed=11, zseed=12, alpha_value=0.2)
    exe = str(tmp_path / 'c_consumer')
    libdir = os.path.join(ROOT, 'gparml_amd')
    subprocess.check_call(['gcc', '-O2', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_consumer.c'), '-o', exe,
                           '-L' + libdir, '-lgparml_hip', '-Wl,-rpath,' + libdir])
    shard = str(tmp_path / 'shard.bin')
    with open(shard, 'wb') as f:
        np.array([N, D, M, Q], dtype=np.int64).tofile(f)
        for a in (d['Y'], d['X_mu'], d['Z'], np.asarray(d['alpha'], dtype=np.float64), np.array([d['sf2'], d['beta']], dtype=np.float64)):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    out = subprocess.run([exe, shard], check=True, capture_output=True, text=True, timeout=300).stdout.splitlines()
    assert out[0].startswith('gparml_hip')
    vals = {}
    for line in out[1:]:
        parts = line.split()
        vals.setdefault(parts[0], []).append(float(parts[-1]))
    got = dict(F=vals['F'][0], grad_sf2=vals['grad_sf2'][0], grad_beta=vals['grad_beta'][0], grad_alpha=np.array(vals['grad_alpha']),
               grad_Z=np.array(vals['grad_Z']).reshape(M, Q))
    eng = ShardEngine(N, D, M, Q)
    eng.upload_shard(d['Y'], d['X_mu'], d['X_S'])
    eng.set_globals(d['Z'], d['sf2'], d['alpha'], d['beta'])
    py = eng.evaluate(False)
    eng.close()
    assert got['F'] == py['F'] and got['grad_sf2'] == py['grad_sf2'] and got['grad_beta'] == py['grad_beta']
    assert np.array_equal(got['grad_Z'], py['grad_Z']) and np.array_equal(got['grad_alpha'], py['grad_alpha'])
    ref = Fz.evaluate(d['Z'], d['sf2'], d['alpha'], d['beta'], d['Y'], d['X_mu'], d['X_S'], want_embeddings=False)
    assert_close(got['F'], ref['F'], F_RTOL, what='F')
    for k in ('grad_Z', 'grad_alpha', 'grad_sf2', 'grad_beta'):
        assert_close(got[k], ref[k], G_RTOL, what=k)
