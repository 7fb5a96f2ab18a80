"""Loader of the HIP C-ABI library.  There is NO fallback: when libomnimamba_hip.so is missing or a tensor is
not on a HIP device every op raises.  (The CPU test-suite injects its emulator build of the same sources through
tests/emu/loader.py -- nothing in this package knows how to find it.)"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libomnimamba_hip.so")
_LIB = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(
            f"omnimamba_amd: HIP extension not built ({path} missing). Run `python -m omnimamba_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for these ops.")
    return _capi.bind(ctypes.CDLL(path))


def get_lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def require_device(lib, *tensors):
    """All present tensors must live where the library executes (a HIP device for the real build)."""
    emu = bool(lib.omk_is_emulated())
    dev = None
    for t in tensors:
        if t is None:
            continue
        if emu:
            if t.is_cuda:
                raise RuntimeError("emulator build only accepts CPU tensors")
        elif not t.is_cuda:
            raise RuntimeError("omnimamba_amd ops run on the MI355X only: got a CPU tensor and there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
