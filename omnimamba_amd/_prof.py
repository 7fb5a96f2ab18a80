"""Optional event-pair timing of individual kernel groups (used by bench.py for the roofline numbers).
Events are recorded on the current stream -- the stream every omk_* kernel is launched on -- and never synchronise."""
from __future__ import annotations

import contextlib

import torch

ENABLED = False
_RANGES = {}
_KERNELS = {}


def note_kernels(name: str, lib):
    """Which kernels the call timed under `name` launched (omk_ssd_last_kernels of the calling thread), last value kept."""
    if ENABLED:
        try:
            _KERNELS[name] = lib.omk_ssd_last_kernels().decode()
        except Exception:      # a foreign library build without the symbol: the bench then reports no kernel id
            _KERNELS[name] = None


def kernels():
    return dict(_KERNELS)


@contextlib.contextmanager
def range_(name: str):
    if not ENABLED or not torch.cuda.is_available():
        yield
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    try:
        yield
    finally:
        e1.record()
        _RANGES.setdefault(name, []).append((e0, e1))


def reset():
    _RANGES.clear()
    _KERNELS.clear()


def summary():
    """name -> (count, mean ms); call after torch.cuda.synchronize()."""
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in _RANGES.items()}
