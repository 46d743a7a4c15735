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
    assert _hip.load().oth_abi_version() == 3
    with pytest.raises(_hip.HipError) as ei:
        _hip.Context(0)
    assert ei.value.code == -2 and 'no CPU fallback' in str(ei.value)
    from ofdm_tools import ofdm_cr_tools as T
    import numpy as np
    with pytest.raises(_hip.HipError):
        T.welch_power_estimate(np.zeros(8192, np.complex64), 4096, 1.0)


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
