"""ctypes mirror of include/omk.h: struct layouts + one thin `call_*` marshaller per entry point.

Nothing here computes: tensors are described by (data_ptr, dtype, shape, element strides) and handed to the
C ABI.  `lib` is always passed in explicitly (omnimamba_amd._lib.get_lib() for the product; the CPU test-suite
passes its emulator build of the same sources).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

OMK_ABI_VERSION = 7
OMK_MAX_DIMS = 5
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.uint8: 3, torch.bool: 3}   # 3 = OMK_U8: masks only


class OmkTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * OMK_MAX_DIMS), ("stride", C.c_int64 * OMK_MAX_DIMS)]


def T(t: Optional[torch.Tensor]) -> OmkTensor:
    o = OmkTensor()
    if t is None:
        return o
    dt = _DT.get(t.dtype)
    if dt is None:
        raise TypeError(f"unsupported dtype {t.dtype} (f32/bf16/f16, u8/bool masks only)")
    n = t.dim()
    if n > OMK_MAX_DIMS:
        raise ValueError("too many dims")
    # (slice assignment: one ctypes call per array instead of two per dimension -- a descriptor is 1.3 us instead of 3.7, and a
    # small launch such as BASELINE configs[0] at batch 2 is bound by the ten of them its wrapper builds)
    o.data, o.dtype, o.ndim = t.data_ptr(), dt, n
    o.shape[:n] = t.shape
    o.stride[:n] = t.stride()
    return o


def _S(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


_t, _f, _i, _i64 = OmkTensor, C.c_float, C.c_int32, C.c_int64
_ws = [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]

AddNormFwd = _S("OmkAddNormFwd", [(n, _t) for n in ("x", "residual", "weight", "bias", "y", "residual_out", "rstd", "mean")]
                + [("eps", _f), ("is_rms_norm", _i)])
AddNormBwd = _S("OmkAddNormBwd", [(n, _t) for n in ("dy", "dresidual_out", "xsum", "weight", "rstd", "mean", "dx",
                                                    "dresidual_in", "dweight", "dbias")] + _ws
                + [("is_rms_norm", _i), ("has_bias", _i)])
NormGatedFwd = _S("OmkNormGatedFwd", [(n, _t) for n in ("x", "z", "weight", "bias", "y", "rstd")]
                  + [("group_size", _i64), ("eps", _f), ("norm_before_gate", _i)])
NormGatedBwd = _S("OmkNormGatedBwd", [(n, _t) for n in ("dy", "x", "z", "weight", "dx", "dz", "dweight")] + _ws
                  + [("group_size", _i64), ("eps", _f), ("norm_before_gate", _i)])
Conv1dFwd = _S("OmkConv1dFwd", [(n, _t) for n in ("x", "weight", "bias", "initial_states", "out", "final_states")]
               + [("silu", _i)])
Conv1dBwd = _S("OmkConv1dBwd", [(n, _t) for n in ("x", "weight", "bias", "initial_states", "dout", "dx", "dweight",
                                                  "dbias", "dinitial_states")] + [("silu", _i), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)])
Conv1dUpdate = _S("OmkConv1dUpdate", [(n, _t) for n in ("x", "conv_state", "weight", "bias", "out")] + [("silu", _i)])
StateUpdate = _S("OmkStateUpdate", [(n, _t) for n in ("state", "x", "dt", "A", "Bm", "Cm", "D", "z", "dt_bias", "out")]
                 + [("dt_softplus", _i)])
SelScanFwd = _S("OmkSelScanFwd", [(n, _t) for n in ("u", "delta", "A", "Bm", "Cm", "D", "z", "delta_bias", "out",
                                                    "last_state", "pass_states")] + [("delta_softplus", _i)])
SelScanBwd = _S("OmkSelScanBwd", [(n, _t) for n in ("u", "delta", "A", "Bm", "Cm", "D", "z", "delta_bias", "dout", "du",
                                                    "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias", "pass_states")] + _ws
                + [("delta_softplus", _i)])
NormLinear = _S("OmkNormLinear", [(n, _t) for n in ("x", "residual", "z", "norm_weight", "weight", "bias", "lora_a", "lora_b",
                                                    "residual_out", "out", "conv_state", "conv_weight", "conv_bias")]
                + [("group_size", C.c_int64), ("conv_offset", C.c_int64), ("eps", _f), ("lora_scale", _f),
                   ("norm_before_gate", _i), ("conv_silu", _i)])
LoraAdd = _S("OmkLoraAdd", [(n, _t) for n in ("out", "h", "lora_b", "mask")] + [("scale", _f)])
LoraUpBwd = _S("OmkLoraUpBwd", [(n, _t) for n in ("dy", "lora_b", "h", "dh", "dlora_b")])
SsdFwd = _S("OmkSsdFwd", [(n, _t) for n in ("x", "dt", "A", "Bm", "Cm", "D", "z", "dt_bias", "initial_states", "out",
                                            "out_x", "final_states", "window_states")] + _ws
            + [("dt_min", _f), ("dt_max", _f), ("dt_softplus", _i), ("chunk_size", _i), ("force_generic", _i), ("flags", _i),
               ("conv_weight", _t), ("conv_bias", _t)])
# OmkSsdFwd.flags / OmkSsdBwd.flags (include/omk.h)
SSD_PRECISE, SSD_KHILO, SSD_EVERY_CHUNK, SSD_NO_SPLIT, SSD_COLUMN_SLICE, SSD_SEQUENTIAL_BWD = 1, 2, 4, 8, 16, 32
SsdBwd = _S("OmkSsdBwd", [(n, _t) for n in ("x", "dt", "A", "Bm", "Cm", "D", "dt_bias", "initial_states", "y", "dout",
                                            "dfinal_states", "dx", "ddt", "dA", "dB", "dC", "dD", "ddt_bias",
                                            "dinitial_states", "window_states")] + _ws
            + [("dt_min", _f), ("dt_max", _f), ("dt_softplus", _i), ("chunk_size", _i), ("force_generic", _i), ("flags", _i)])

CrossEntropy = _S("OmkCrossEntropy", [("logits", _t), ("labels", C.c_void_p), ("losses", _t), ("grad_scale", C.c_void_p),
                                      ("ignore_index", _i64), ("write_grad", _i)])

Sample = _S("OmkSample", [("logits", _t), ("out_ids", _t), ("step_counter", C.c_void_p), ("seed", C.c_uint64), ("offset", C.c_uint64),
                          ("top_k", _i), ("top_p", _f), ("temperature", _f), ("min_p", _f)])

STRUCTS = {s.__name__: s for s in (Sample, CrossEntropy, OmkTensor, AddNormFwd, AddNormBwd, NormGatedFwd, NormGatedBwd, Conv1dFwd, Conv1dBwd,
                                   Conv1dUpdate, StateUpdate, SelScanFwd, SelScanBwd, NormLinear, LoraAdd, LoraUpBwd, SsdFwd, SsdBwd)}

# every symbol include/omk.h declares
SYMBOLS = [
    "omk_abi_version", "omk_last_error", "omk_is_emulated", "omk_sizeof", "omk_ssd_last_kernels",
    "omk_add_norm_fwd", "omk_add_norm_bwd_workspace_bytes", "omk_add_norm_bwd",
    "omk_norm_gated_fwd", "omk_norm_gated_bwd_workspace_bytes", "omk_norm_gated_bwd",
    "omk_causal_conv1d_fwd", "omk_causal_conv1d_bwd_workspace_bytes", "omk_causal_conv1d_bwd", "omk_causal_conv1d_update",
    "omk_selective_state_update", "omk_norm_linear", "omk_lora_add", "omk_lora_up_bwd",
    "omk_selective_scan_fwd", "omk_selective_scan_fwd_form", "omk_selective_scan_bwd_form", "omk_selective_scan_bwd_workspace_bytes", "omk_selective_scan_bwd",
    "omk_ssd_scan_fwd_workspace_bytes", "omk_ssd_scan_fwd_window_states_bytes", "omk_ssd_scan_fwd", "omk_ssd_scan_bwd_workspace_bytes", "omk_ssd_scan_bwd",
    "omk_cross_entropy", "omk_lora_up_bwd_parts", "omk_sample",
]


def bind(lib: C.CDLL) -> C.CDLL:
    """Declare prototypes on a freshly dlopen'ed library and check ABI version + struct sizes."""
    lib.omk_abi_version.restype = C.c_int
    lib.omk_last_error.restype = C.c_char_p
    lib.omk_ssd_last_kernels.restype = C.c_char_p
    lib.omk_is_emulated.restype = C.c_int
    lib.omk_sizeof.restype = C.c_size_t
    lib.omk_sizeof.argtypes = [C.c_char_p]
    for s in SYMBOLS:
        fn = getattr(lib, s)  # raises AttributeError when a declared symbol is missing
        if s.endswith("_workspace_bytes") or s.endswith("_window_states_bytes"):
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p]
        elif s in ("omk_selective_scan_fwd_form", "omk_selective_scan_bwd_form"):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p]
        elif s == "omk_lora_up_bwd_parts":
            fn.restype = C.c_int
            fn.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
        elif s not in ("omk_abi_version", "omk_last_error", "omk_is_emulated", "omk_sizeof", "omk_ssd_last_kernels"):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p]
    if lib.omk_abi_version() != OMK_ABI_VERSION:
        raise RuntimeError(f"libomnimamba_hip ABI {lib.omk_abi_version()} != python {OMK_ABI_VERSION}")
    for name, s in STRUCTS.items():
        n = lib.omk_sizeof(name.encode())
        if n != C.sizeof(s):
            raise RuntimeError(f"struct {name}: C sizeof {n} != ctypes {C.sizeof(s)}")
    return lib


def check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.omk_last_error()
        raise RuntimeError(f"{what} failed (omk_status {rc}): {msg.decode() if msg else ''}")


def stream_of(lib, t: torch.Tensor):
    if lib.omk_is_emulated():
        return None
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def run(lib, fn_name: str, params, ref: torch.Tensor):
    """Call `fn_name(&params, stream)` with the current stream of `ref`'s device selected."""
    fn = getattr(lib, fn_name)
    if lib.omk_is_emulated():
        check(lib, fn(C.byref(params), None), fn_name)
    else:
        with torch.cuda.device(ref.device):
            check(lib, fn(C.byref(params), stream_of(lib, ref)), fn_name)


def workspace(lib, fn_name: str, params, ref: torch.Tensor) -> Optional[torch.Tensor]:
    n = getattr(lib, fn_name)(C.byref(params))
    if n == 0:
        return None
    ws = torch.empty(n, dtype=torch.uint8, device=ref.device)
    params.workspace = ws.data_ptr()
    params.workspace_bytes = n
    return ws
