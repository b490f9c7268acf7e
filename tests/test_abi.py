"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly the symbols that
include/gparml_hip.h declares; the Python binding knows each of them; no compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'gparml_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gp_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_path():
    syms = _header_symbols()
    for needed in ('gp_create', 'gp_destroy', 'gp_upload_shard', 'gp_set_globals', 'gp_phase1', 'gp_stats_buffer',
                   'gp_global_step', 'gp_phase2', 'gp_grads_buffer', 'gp_finish', 'gp_download', 'gp_last_error'):
        assert needed in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, 'gparml_amd', 'libgparml_hip.so'))
    for s in _header_symbols():
        assert hasattr(lib, s), 'library does not export %s' % s


def test_library_exports_nothing_the_header_does_not_declare():
    """header symbols == exported gp_* symbols: an entry point that is not in include/gparml_hip.h is either declared there or made static."""
    so = os.path.join(ROOT, 'gparml_amd', 'libgparml_hip.so')
    out = subprocess.run(['nm', '-D', '--defined-only', so], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in 'TW' and ln.split()[-1].startswith('gp_')})
    assert exported == _header_symbols(), (sorted(set(exported) - set(_header_symbols())), sorted(set(_header_symbols()) - set(exported)))


def test_binding_covers_every_declared_symbol():
    from gparml_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    lib = _lib.load()
    assert lib.gp_version().decode().startswith('gparml_hip')


def test_no_product_import_of_the_oracle():
    """The product path must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, 'gparml_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(dirpath, f)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from gparml_amd.engine import ShardEngine
    from gparml_amd._lib import GparmlHipError
    try:
        ShardEngine(10, 2, 3, 2)
    except GparmlHipError as e:
        assert 'HIP' in str(e) or 'device' in str(e)
    else:
        raise AssertionError('ShardEngine must raise without a GPU (no CPU fallback)')
