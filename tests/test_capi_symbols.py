"""The C-ABI library builds for gfx950, loads, and exports every symbol include/catint_pnp.h
declares (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from catint_amd.build import build_library
    path = build_library()
    return ctypes.CDLL(path)


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'catint_pnp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pnp_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    from catint_amd import _capi
    assert declared_symbols() == sorted(_capi.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), s


def test_version_and_config_abi(lib):
    from catint_amd import _capi
    lib.pnp_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.pnp_version()
    assert ctypes.sizeof(_capi.PnpConfig) == 8 * 4 + 8 + 4 * 8


def test_create_rejects_bad_config_without_gpu(lib):
    # argument validation happens before any device call, so it is testable on CPU
    from catint_amd import _capi
    L = _capi.load_library()
    h = ctypes.c_void_p()
    cfg = _capi.PnpConfig(ctypes.sizeof(_capi.PnpConfig), 0, 3, 5000, 0, 0, 0, 1, 4, 1e-10, 1e-12, 4e-4, 7e-10)
    assert L.pnp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b'nx' in L.pnp_last_error(None)
    cfg = _capi.PnpConfig(ctypes.sizeof(_capi.PnpConfig), 0, 3, 100, 7, 0, 0, 1, 4, 1e-10, 1e-12, 4e-4, 7e-10)
    assert L.pnp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b'calculator' in L.pnp_last_error(None)
    cfg = _capi.PnpConfig(ctypes.sizeof(_capi.PnpConfig), 0, 3, 100, 0, 0, 0, 0, 4, 1e-10, 1e-12, 4e-4, 7e-10)
    assert L.pnp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b'use_migration' in L.pnp_last_error(None)
    cfg = _capi.PnpConfig(12, 0, 3, 100, 0, 0, 0, 1, 4, 1e-10, 1e-12, 4e-4, 7e-10)
    assert L.pnp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from catint_amd import _capi
    monkeypatch.setattr(_capi, '_lib', None)
    monkeypatch.setattr(_capi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_capi.PnpLibraryError):
        _capi.load_library()
