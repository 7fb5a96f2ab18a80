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


def test_last_kernels_names_what_a_scan_call_launched():
    """omk_ssd_last_kernels (ABI 6): the measurement aid bench.py uses to tie a PMC traffic file to the kernels it timed.  On the emulator
    the same host code runs: a plain bf16 forward of the class A shape reports the dt' preparation and the specialised-wave kernel
    with its template arguments, a PRECISE call the PRECISE instantiation, a training forward the one that writes window states."""
    import torch
    from emu.loader import use_emulator
    with use_emulator():
        import omnimamba_amd.ssd_combined as S
        from omnimamba_amd import _capi as K
        from omnimamba_amd._lib import get_lib
        lib = get_lib()
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 70, 2, 64, generator=g).bfloat16()
        dt = torch.randn(1, 70, 2, generator=g).bfloat16()
        A = -torch.rand(2, generator=g) - 0.5
        Bm, Cm = torch.randn(1, 70, 1, 128, generator=g).bfloat16(), torch.randn(1, 70, 1, 128, generator=g).bfloat16()
        S.ssd_scan_fwd(x, dt, A, Bm, Cm, dt_softplus=True)
        assert lib.omk_ssd_last_kernels().decode() == "ssd_dt_prep;ssd_a8<mode=0,dump=0,khilo=0,precise=0>"
        S.ssd_scan_fwd(x, dt, A, Bm, Cm, dt_softplus=True, flags=K.SSD_PRECISE)
        assert lib.omk_ssd_last_kernels().decode() == "ssd_dt_prep;ssd_a8<mode=0,dump=0,khilo=1,precise=1>"
        r = S.ssd_scan_fwd(x, dt, A, Bm, Cm, dt_softplus=True, save_window_states=True)
        assert r[3] is not None and lib.omk_ssd_last_kernels().decode() == "ssd_dt_prep;ssd_a8<mode=0,dump=1,khilo=0,precise=0>"
        S.ssd_scan_fwd(x, dt, A, Bm, Cm, dt_softplus=True, flags=K.SSD_COLUMN_SLICE)
        assert lib.omk_ssd_last_kernels().decode().startswith("ssd_dt_prep;ssd_a6<mode=0")
