"""The C-ABI library loads on a GPU-less host and exports every symbol include/dcvc_amd.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from opendcvc_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "dcvc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcvc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdcvc_amd.so not built (run __graft_entry__.build())")
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) > 50
    for n in names:
        assert hasattr(L, n), f"{n} declared in dcvc_amd.h but not exported"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)


def test_loader_binds_and_reports_version():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdcvc_amd.so not built")
    L = _lib.lib()
    assert L.dcvc_abi_version() == 1
    assert isinstance(L.dcvc_last_error(), bytes)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.DcvcError):
        _lib.lib()
