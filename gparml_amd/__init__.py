"""gparml_amd -- MI355X (gfx950) implementation of GParML's per-shard ``partial_terms`` hot path.

The compute lives in ``libgparml_hip.so`` (hand-written HIP, built from ``gparml_amd/csrc``; C ABI in
``include/gparml_hip.h``).  This package is the thin Python host side that mirrors the reference's own
interfaces for the path:

* ``gparml_amd.partial_terms.partial_terms``  -- the class of partial_terms.py:15
* ``gparml_amd.gpu_MapReduce``                 -- a MapReduce backend module with the function set of
  local_MapReduce.py (init / cache / statistics_MR / embeddings_MR / load / save / ...)
* ``gparml_amd.engine.ShardEngine``            -- one shard resident on one GPU (two-phase evaluation)
* ``gparml_amd.dist``                          -- one process per GPU, all-reduce of the partial sums (RCCL)

There is no CPU fallback: importing the engine without the built library, or using it without a GPU,
raises.
"""
__version__ = '0.1.0'
