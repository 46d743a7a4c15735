"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/ofdm_tools_hip.h declares, and refuses to compute without a GPU
(no silent fallback).  No compute calls are made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'ofdm_tools_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(oth_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ('oth_ctx_create', 'oth_welch_plan', 'oth_welch_exec', 'oth_welch_accumulate',
                 'oth_welch_finalize', 'oth_csd_exec', 'oth_chain_push', 'oth_channel_power', 'oth_xcorr',
                 'oth_last_error'):
        assert must in names
    assert len(names) >= 40


def test_library_exports_every_declared_symbol():
    from ofdm_tools import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip('library not built yet (run __graft_entry__.build())')
    lib = ctypes.CDLL(_hip.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from ofdm_tools import _hip
    assert sorted(_hip.SIGNATURES) == declared_symbols()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ofdm_tools import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip('library not built yet')
    assert _hip.load().oth_abi_version() == 5
    with pytest.raises(_hip.HipError) as ei:
        _hip.Context(0)
    assert ei.value.code == -2 and 'no CPU fallback' in str(ei.value)
    from ofdm_tools import ofdm_cr_tools as T
    import numpy as np
    with pytest.raises(_hip.HipError):
        T.welch_power_estimate(np.zeros(8192, np.complex64), 4096, 1.0)


def test_exception_barrier_at_the_abi():
    """include/ofdm_tools_hip.h: "nothing throws or aborts".  A C++ exception below an entry point must come back as
    an error code (a bad_alloc crossing ctypes would be std::terminate and take the flowgraph down).  The
    oth__debug_throw hook raises inside the same OTH_TRY / OTH_CATCH pair every entry point has; no GPU needed."""
    from ofdm_tools import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip('library not built yet')
    lib = ctypes.CDLL(_hip.LIB_PATH)
    lib.oth__debug_throw.restype = ctypes.c_int
    lib.oth__debug_throw.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.oth_last_error.restype = ctypes.c_char_p
    lib.oth_last_error.argtypes = [ctypes.c_void_p]
    lib.oth_strerror.restype = ctypes.c_char_p
    assert lib.oth__debug_throw(None, -1) == 0
    assert lib.oth__debug_throw(None, 0) == -4 and b'memory' in lib.oth_last_error(None)
    assert lib.oth__debug_throw(None, 1) == -6 and lib.oth_last_error(None) == b'debug: runtime_error'
    assert lib.oth__debug_throw(None, 2) == -6 and b'unknown C++ exception' in lib.oth_last_error(None)
    assert lib.oth__debug_throw(None, 3) in (-4, -6)             # a real over-sized std::vector, not a staged throw
    assert b'internal' in lib.oth_strerror(-6)
    # and every extern "C" body in the source sits inside the barrier
    src = open(os.path.join(ROOT, 'gr-ofdm_tools_amd', 'csrc', 'api.hip')).read()
    ext = src[src.index('extern "C" {'):]
    bodies = re.findall(r'^int (oth_\w+)\([^)]*\) \{\n(.*?)^\}', ext, flags=re.S | re.M)
    assert len(bodies) >= 55
    for name, body in bodies:
        assert body.lstrip().startswith('OTH_TRY') and 'OTH_CATCH(' in body.rstrip().splitlines()[-1], name


def test_bench_gpus_n_without_a_gpu_exits_with_the_clear_message():
    """`python bench.py --gpus N` is the driver's command shape: without a launcher it must decide BEFORE touching the
    GPU whether it can spawn its ranks, and say why not."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip('GPU present')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    for extra in ([], ['--gpus', '2']):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1'] + extra, env=env,
                           capture_output=True, timeout=300)
        assert p.returncode != 0 and b'needs an MI355X' in p.stderr and b'no CPU fallback' in p.stderr, p.stderr
        assert p.stdout.strip() == b''


def test_missing_library_fails_loudly(monkeypatch):
    from ofdm_tools import _hip
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', '/nonexistent/libofdmtools_hip.so')
    with pytest.raises(_hip.HipUnavailable):
        _hip.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gr-ofdm_tools_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.replace('oracle/', ''), f
                assert 'import scipy' not in text and 'from scipy' not in text, f
