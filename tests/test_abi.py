"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/caiman_rnnt.h declares; argument validation returns errors without touching a GPU."""
import ctypes
import os

import pytest

from caiman_asr_amd import _lib


@pytest.fixture(scope="module")
def native():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.lib()


def test_header_symbols_are_exported(native):
    names = _lib.exported_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(native, n), f"{n} declared in include/caiman_rnnt.h but not exported"
    # and every bound signature is declared in the header
    for n in _lib._SIGS:
        assert n in names, f"{n} bound in _lib.py but missing from the header"


def test_identity(native):
    assert native.caiman_abi_version() >= 1
    assert native.caiman_built_for_gfx950() == 1


def test_code_object_targets_gfx950():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80"):
        assert other not in blob


def test_argument_errors_without_gpu(native):
    # blank index out of range -> CAIMAN_ERR_INVALID with the reference's message
    # (training/lib/csrc/transducer_loss.cu:429-434); validation happens before any HIP call.
    one = ctypes.c_void_p(16)
    rc = native.caiman_transducer_loss_forward(one, one, one, one, one, one, 1, 1, 1, 4, 0.0, 7, 0.0, -1,
                                               0.0, -2, 0, 1, one, one, one, None)
    assert rc == 1
    assert b"Expected blank index to be in the range of 0 to 3, but got 7" in native.caiman_last_error()
    rc = native.caiman_logsumexp(one, 2, 8, 4, 1, one, 1, 128, None)
    assert rc == 1 and b"alias" in native.caiman_last_error()
    rc = native.caiman_logsumexp(one, 2, 8, 8, 9, one, 1, 128, None)
    assert rc == 3 and b"dtype" in native.caiman_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MissingNativeLibrary):
        _lib.lib()
