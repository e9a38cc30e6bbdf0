"""ctypes binding of the C ABI in include/qsae.h (libqsae_hip.so).

This is the only way the package reaches the GPU kernels; there is no CPU or eager-PyTorch
fallback.  If the library is missing or fails to load, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

_PKG = Path(__file__).resolve().parent
import os

# QSAE_HIP_LIB: another build of the same library (same-box A/B timing of two builds); default: the in-tree one
LIB_PATH = Path(os.environ["QSAE_HIP_LIB"]) if os.environ.get("QSAE_HIP_LIB") else _PKG / "lib" / "libqsae_hip.so"
# The debug build (same sources + -DQSAE_DEBUG_BUILD): process-wide qsae_debug_* switches and ablation kernels.  Never
# loaded by the package itself; tools/ and a few tests ask for it explicitly (use_library("debug")).
DEBUG_LIB_PATH = _PKG / "lib" / "libqsae_hip_debug.so"
ABI_VERSION = 4

OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/qsae.h
SIGNATURES = {
    "qsae_abi_version": (_i, []),
    "qsae_last_error": (C.c_char_p, []),
    "qsae_device_info": (_i, [C.POINTER(_i), C.c_char_p, _i]),
    "qsae_profile_sweep_events": (_i, [_vp, _vp]),
    "qsae_profile_event_create": (_i, [C.POINTER(_vp)]),
    "qsae_profile_event_destroy": (_i, [_vp]),
    "qsae_profile_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "qsae_profile_sweep_flop_fraction": (C.c_double, [_i]),
    "qsae_kperm_rows": (_i, [_vp, _i, _i, _vp, _vp]),
    "qsae_encode_dense": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp]),
    "qsae_encode_dense_kperm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp]),
    "qsae_emu_w_bytes": (_sz, [_i, _i]),
    "qsae_emu_pack_w": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "qsae_encode_dense_emu_workspace_bytes": (_sz, [_i, _i]),
    "qsae_encode_dense_emu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "qsae_encode_bits": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "qsae_topk_rows": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp]),
    "qsae_encode_topk_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "qsae_encode_topk": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "qsae_encode_topk_kperm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "qsae_encode_topk_latent": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "qsae_prefilter_w_bytes": (_sz, [_i, _i]),
    "qsae_prefilter_pack_w": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "qsae_encode_topk_prefilter_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "qsae_encode_topk_prefilter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _sz, _i,
                                         C.POINTER(_i), _vp]),
    "qsae_binary_forward_prefilter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                            _vp, _sz, _i, C.POINTER(_i), _vp]),
    "qsae_prefilter_submit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                    _vp, _sz, _vp, _vp]),
    "qsae_prefilter_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                    _vp, _sz, _i, _vp]),
    "qsae_table_forward_prefilter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                           _vp, _sz, _i, C.POINTER(_i), _vp]),
    "qsae_prefilter_submit_table": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                          _vp, _sz, _vp, _vp]),
    "qsae_prefilter_finish_table": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _i64, _vp,
                                          _vp, _sz, _i, _vp]),
    "qsae_densify": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "qsae_binary_row_bytes": (_i, [_i, _i]),
    "qsae_pack_binary": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "qsae_unpack_binary": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "qsae_decode_binary_sparse": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "qsae_decode_table_sparse": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "qsae_binary_soft_table": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "qsae_pack_ternary": (_i, [_vp, _i, _i, _vp, _vp]),
    "qsae_decode_ternary_dense": (_i, [_vp, _i64, _i, _i, _vp, _i, _vp, _vp]),
    "qsae_matryoshka_sizes": (_i, [_i, _i, _vp]),
    "qsae_pack_matryoshka": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "qsae_decode_matryoshka": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "qsae_encode_bits_prefilter_workspace_bytes": (_sz, [_i, _i, _i]),
    "qsae_encode_bits_prefilter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, C.POINTER(_i), _vp]),
    "qsae_encode_bits_prefilter_submit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, _vp, _vp]),
    "qsae_encode_bits_prefilter_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, _i, _vp]),
    "qsae_encode_bits_band_workspace_bytes": (_sz, [_i, _i, _i]),
    "qsae_encode_bits_band": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, C.POINTER(_i), _vp]),
    "qsae_encode_bits_band_submit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, _vp, _vp]),
    "qsae_encode_bits_band_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _sz, _i, _vp]),
    "qsae_pack_matryoshka_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "qsae_decode_matryoshka_sparse": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "qsae_split_dec_supported": (_i, [_i, _i, _i]),
    "qsae_expand_codes_bf16_bytes": (_sz, [_i, _i]),
    "qsae_expand_codes_bf16": (_i, [_vp, _i, _i, _vp, _vp]),
    "qsae_decode_ternary_dense_split": (_i, [_vp, _i64, _i, _i, _vp, _i, _vp, _vp]),
    "qsae_split_scale_bf16": (_i, [_vp, _i, _vp, _vp]),
    "qsae_decode_matryoshka_split": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "qsae_pack_bits_gt": (_i, [_vp, _i64, _i, _i, _f, _vp, _i64, _vp]),
    "qsae_residual_update": (_i, [_vp, _vp, _sz, _f, _vp, _vp]),
    "qsae_threshold_ge": (_i, [_vp, _sz, _f, _vp, _vp]),
    "qsae_scale_bias_rows": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp]),
    "qsae_sq_err_sum": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "qsae_activation_counts": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "qsae_activation_counts_bits": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "qsae_coactivation_sparse": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "qsae_quantize_bits": (_i, [_vp, _i64, _i, _i, _i, _f, _i, _vp, _vp]),
}


class QsaeError(RuntimeError):
    """A libqsae_hip call returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libqsae_hip error {code}: {message}")
        self.code = code


_libs = {}          # "product" / "debug" -> CDLL
_active = "product"


def _open(path: Path) -> C.CDLL:
    if not path.exists():
        raise RuntimeError(
            f"{path} is missing: build it with `python -m quantizedsae_amd.build` "
            "(hipcc --offload-arch=gfx950).  quantizedsae_amd has no CPU / eager fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here means the library is stale
        fn.restype = res
        fn.argtypes = args
    if lib.qsae_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path.name}: ABI version mismatch; rebuild it")
    return lib


def load() -> C.CDLL:
    """The library the package computes with: libqsae_hip.so (loaded once).  Raises RuntimeError if it is missing --
    no fallback.  Inside a `use_library("debug")` block (tests / tools only) it is the debug build instead."""
    lib = _libs.get(_active)
    if lib is None:
        lib = _libs[_active] = _open(LIB_PATH if _active == "product" else DEBUG_LIB_PATH)
    return lib


class use_library:
    """Context manager for tests and tools: route the package's calls to the debug build (qsae_debug_* switches,
    ablation kernels) for the duration of the block.  Not thread-safe, not for product code."""

    def __init__(self, which: str):
        if which not in ("product", "debug"):
            raise ValueError(which)
        self.which = which

    def __enter__(self):
        global _active
        self.prev, _active = _active, self.which
        return load()

    def __exit__(self, *exc):
        global _active
        _active = self.prev
        return False


def check(code: int) -> None:
    if code != OK:
        msg = load().qsae_last_error()
        raise QsaeError(code, msg.decode(errors="replace") if msg else "")
