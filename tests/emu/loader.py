"""Test-only: build + load the emulator variant of the kernels and inject it into omnimamba_amd._lib so the whole
Python stack (marshalling, autograd Functions, modules) can be exercised on CPU tensors."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        from omnimamba_amd import _capi
        path = build_emu.build()
        _EMU = _capi.bind(ctypes.CDLL(path))
        assert _EMU.omk_is_emulated() == 1
    return _EMU


class use_emulator:
    """with use_emulator(): ...   -- omnimamba_amd ops run on CPU tensors through the emulated kernels."""

    def __enter__(self):
        import omnimamba_amd._lib as L
        self._prev = L._LIB
        L._LIB = emu_lib()
        return L._LIB

    def __exit__(self, *a):
        import omnimamba_amd._lib as L
        L._LIB = self._prev
