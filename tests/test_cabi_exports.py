"""CPU: the C-ABI library loads and exports every symbol include/houv_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "houv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(houv_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from houv_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = _declared()
    assert "houv_chamfer_forward" in names and "houv_solve_iterate" in names and "houv_kabsch" in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/houv_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names          # the Python binding covers the whole header
    assert _lib.load().houv_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from houv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HouvHipError):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch
    from houv_amd import _lib
    from houv_amd.metrics import cd
    with pytest.raises(_lib.HouvHipError):
        cd()(torch.rand(1, 4, 3), torch.rand(1, 5, 3))
