"""The C-ABI library: loads, exports every symbol include/omk.h declares, struct layouts agree with ctypes.
No compute (runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "omk.h")).read()
    return sorted(set(re.findall(r"\b(omk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_listed_in_python_mirror():
    from omnimamba_amd import _capi
    assert _declared_symbols() == sorted(_capi.SYMBOLS)


def test_hip_library_exports_everything():
    from omnimamba_amd import _capi, _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _capi.bind(ctypes.CDLL(_lib.LIB_PATH))     # bind() checks ABI version + every struct size
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    assert lib.omk_is_emulated() == 0
    assert lib.omk_sizeof(b"OmkTensor") == ctypes.sizeof(_capi.OmkTensor)


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: with the real library loaded, CPU tensors are rejected loudly."""
    import torch
    from omnimamba_amd import _lib
    from omnimamba_amd.layer_norm import rms_norm_fn
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    _lib._LIB = None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rms_norm_fn(torch.randn(2, 8), torch.ones(8), None)
