"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the public
header declares, and reports argument errors as status codes + messages (no GPU needed)."""

import ctypes as C

import pytest

from speech_to_speech_amd import native


def test_library_loads_and_reports_version():
    lib = native.load()
    assert lib.s2s_abi_version() == 5
    assert lib.s2s_device_count() >= 0


def test_every_declared_symbol_is_exported():
    lib = native.load()
    declared = native.declared_symbols() + native.declared_symbols(native.DEV_HEADER_PATH)
    assert len(declared) >= 10
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []


def test_every_declared_symbol_has_a_binding():
    declared = set(native.declared_symbols())
    bound = set(native._SIGNATURES)
    assert declared - bound == set(), "header symbols without ctypes signature"
    assert bound - declared == set(), "ctypes signatures for symbols missing from the header"
    # the development / measurement hooks live in their own header and binding table: none of them leaks into the public boundary
    dev_declared, dev_bound = set(native.declared_symbols(native.DEV_HEADER_PATH)), set(native._DEV_SIGNATURES)
    assert dev_declared == dev_bound and not (dev_declared & declared)
    assert not any(n.endswith(("_tune", "_debug")) or "graph_bench" in n or "s2s_prof" in n for n in declared)


def test_argument_errors_are_status_codes_not_crashes():
    lib = native.load()
    handle = C.c_void_p()
    st = lib.s2s_frontend_create(7, 80, None, 4, C.byref(handle))
    assert st == -1
    assert b"frontend kind" in lib.s2s_last_error()
    with pytest.raises(native.NativeError):
        native.check(st, "s2s_frontend_create")
    # null library handle on an op entry point
    assert lib.s2s_logmel_whisper(None, None, None, 0, None, 1, None, 0) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        native.load()
