"""CPU oracle for the GParML `partial_terms` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gparml_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

Parity status: PINNED.  ``oracle/literal.py`` is checked against golden vectors captured
from the imported reference (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``)
by ``tests/test_oracle_golden.py``; ``oracle/factorised.py`` (the two-phase formulation the
HIP kernels implement, and the fair multi-core CPU baseline) is checked against
``oracle/literal.py`` and the same golden vectors.
"""
