"""``torch.ops.qsae.*``: the kernels of libqsae_hip.so registered with the PyTorch dispatcher.

The reference's compute is a sequence of ATen ops (``F.linear``, ``torch.topk``, ``scatter_``, ``matmul``: sae/binary.py:24-47,
91-103, sae/baseline.py:17-40, sae/ternary.py:41-52, sae/quantized_matryoshka.py:47-143); here every kernel entry point of the
C ABI (include/qsae.h) is a dispatcher op of namespace ``qsae`` with a fake (meta) implementation, so the module classes'
forwards are visible to ``torch.compile`` / ``torch.export`` as graph nodes instead of opaque ctypes calls.  Two layers:

* kernel level -- ``torch.ops.qsae.encode_dense``, ``.encode_topk_prefilter``, ``.binary_forward_prefilter``,
  ``.decode_binary_sparse`` ...: thin registrations over ``quantizedsae_amd.ops`` (which binds the C ABI with ctypes; torch only
  supplies device memory and the current stream).  The module classes call these (through the same-named Python functions of
  this module, which keep the keyword defaults of ``ops.py``).
* model level -- ``torch.ops.qsae.binary_sae_forward`` / ``baseline_sae_forward`` / ``ternary_sae_forward`` /
  ``matryoshka_sae_forward`` / ``residual_sae_forward``: one node per ``forward()``, used by the module classes while they are
  being traced (``torch.compiler.is_compiling()``).  The Python around the kernels (derived-weight caches keyed on parameter
  versions, path selection by shape, the host read of the flagged-row count) is not traceable and does not need to be: the node
  takes the input batch and the parameters and finds its module through a process-local handle.

CUDA (ROCm) tensors only: there is no CPU implementation to register, the ops raise on CPU tensors like ``ops.py`` does.
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops as _ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID  # noqa: F401

_NS = "qsae"


def _op(name, mutates=()):
    return torch.library.custom_op(f"{_NS}::{name}", mutates_args=mutates)


def _i32(shape, like):
    return torch.empty(shape, dtype=torch.int32, device=like.device)


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _host_count():
    return torch.empty((), dtype=torch.int64)


# ---- encoder -------------------------------------------------------------------------------------------------------
@_op("kperm_rows")
def _kperm_rows(src: Tensor) -> Tensor:
    return _ops.kperm_rows(src)


@_kperm_rows.register_fake
def _(src):
    return _f32(src.shape, src)


@_op("encode_dense")
def _encode_dense(x: Tensor, W: Tensor, bias: Optional[Tensor], act: int, kperm: bool) -> Tensor:
    return _ops.encode_dense(x, W, bias, act, kperm=kperm)


@_encode_dense.register_fake
def _(x, W, bias, act, kperm):
    return _f32((x.shape[0], W.shape[0]), x)


@_op("emu_pack_w")
def _emu_pack_w(W: Tensor) -> Tuple[Tensor, Tensor]:
    return _ops.emu_pack_w(W)


@_emu_pack_w.register_fake
def _(W):
    return torch.empty((W.shape[0], 3 * W.shape[1]), dtype=torch.float16, device=W.device), _f32((2,), W)


@_op("encode_dense_emu")
def _encode_dense_emu(x: Tensor, Wc: Tensor, meta2: Tensor, bias: Optional[Tensor], act: int) -> Tensor:
    return _ops.encode_dense_emu(x, Wc, meta2, bias, act)


@_encode_dense_emu.register_fake
def _(x, Wc, meta2, bias, act):
    return _f32((x.shape[0], Wc.shape[0]), x)


@_op("encode_bits")
def _encode_bits(x: Tensor, W: Tensor, bias: Optional[Tensor]) -> Tensor:
    return _ops.encode_bits(x, W, bias)


@_encode_bits.register_fake
def _(x, W, bias):
    return _i32((x.shape[0], (W.shape[0] + 31) // 32), x)


@_op("encode_bits_prefilter")
def _encode_bits_prefilter(x: Tensor, W: Tensor, bias: Optional[Tensor], Wq: Tensor, meta: Tensor) -> Tuple[Tensor, Tensor]:
    z, flagged = _ops.encode_bits_prefilter(x, W, bias, Wq, meta)
    return z, torch.tensor(flagged, dtype=torch.int64)


@_encode_bits_prefilter.register_fake
def _(x, W, bias, Wq, meta):
    return _i32((x.shape[0], (W.shape[0] + 31) // 32), x), _host_count()


@_op("encode_bits_band")
def _encode_bits_band(x: Tensor, W: Tensor, bias: Optional[Tensor], Wq: Tensor, meta: Tensor) -> Tuple[Tensor, Tensor]:
    z, flagged = _ops.encode_bits_band(x, W, bias, Wq, meta)
    return z, torch.tensor(flagged, dtype=torch.int64)


@_encode_bits_band.register_fake
def _(x, W, bias, Wq, meta):
    return _i32((x.shape[0], (W.shape[0] + 31) // 32), x), _host_count()


@_op("topk_rows", mutates=("latent",))
def _topk_rows(latent: Tensor, k: int, zero_rest: bool) -> Tuple[Tensor, Tensor]:
    return _ops.topk_rows(latent, k, zero_rest)


@_topk_rows.register_fake
def _(latent, k, zero_rest):
    return _i32((latent.shape[0], k), latent), _f32((latent.shape[0], k), latent)


@_op("encode_topk")
def _encode_topk(x: Tensor, W: Tensor, bias: Optional[Tensor], k: int, kperm: bool) -> Tuple[Tensor, Tensor]:
    return _ops.encode_topk(x, W, bias, k, kperm=kperm)


@_encode_topk.register_fake
def _(x, W, bias, k, kperm):
    return _i32((x.shape[0], k), x), _f32((x.shape[0], k), x)


@_op("encode_topk_latent")
def _encode_topk_latent(x: Tensor, W: Tensor, bias: Optional[Tensor], k: int, kperm: bool) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops.encode_topk_latent(x, W, bias, k, kperm=kperm)


@_encode_topk_latent.register_fake
def _(x, W, bias, k, kperm):
    return _i32((x.shape[0], k), x), _f32((x.shape[0], k), x), _f32((x.shape[0], W.shape[0]), x)


@_op("prefilter_pack_w")
def _prefilter_pack_w(W: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    return _ops.prefilter_pack_w(W, bias)


@_prefilter_pack_w.register_fake
def _(W, bias):
    return torch.empty(W.shape, dtype=torch.float16, device=W.device), _f32((4,), W)


def _dense_or_empty(dense, like, H):
    return dense if dense is not None else _f32((0, H), like)


@_op("encode_topk_prefilter")
def _encode_topk_prefilter(x: Tensor, W: Tensor, bias: Optional[Tensor], Wq: Tensor, meta: Tensor, k: int, want_dense: bool,
                           spec_rows: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    info = {}
    idx, val, dense = _ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=want_dense, spec_rows=spec_rows, info=info)
    return idx, val, _dense_or_empty(dense, x, W.shape[0]), torch.tensor(info["flagged_rows"], dtype=torch.int64)


@_encode_topk_prefilter.register_fake
def _(x, W, bias, Wq, meta, k, want_dense, spec_rows):
    B, H = x.shape[0], W.shape[0]
    return _i32((B, k), x), _f32((B, k), x), _f32((B if want_dense else 0, H), x), _host_count()


@_op("binary_forward_prefilter")
def _binary_forward_prefilter(x: Tensor, W: Tensor, bias: Optional[Tensor], Wq: Tensor, meta: Tensor, k: int, packed: Tensor,
                              n_bits: int, step: float, dec_bias: Optional[Tensor], want_dense: bool,
                              spec_rows: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    info = {}
    idx, val, dense, recon = _ops.binary_forward_prefilter(x, W, bias, Wq, meta, k, packed, n_bits, step, dec_bias,
                                                           want_dense=want_dense, spec_rows=spec_rows, info=info)
    return idx, val, _dense_or_empty(dense, x, W.shape[0]), recon, torch.tensor(info["flagged_rows"], dtype=torch.int64)


@_binary_forward_prefilter.register_fake
def _(x, W, bias, Wq, meta, k, packed, n_bits, step, dec_bias, want_dense, spec_rows):
    B, H = x.shape[0], W.shape[0]
    return _i32((B, k), x), _f32((B, k), x), _f32((B if want_dense else 0, H), x), _f32((B, x.shape[1]), x), _host_count()


@_op("table_forward_prefilter")
def _table_forward_prefilter(x: Tensor, W: Tensor, bias: Optional[Tensor], Wq: Tensor, meta: Tensor, k: int, table: Tensor,
                             scale: float, dec_bias: Optional[Tensor], want_dense: bool,
                             spec_rows: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    info = {}
    idx, val, dense, recon = _ops.table_forward_prefilter(x, W, bias, Wq, meta, k, table, scale, dec_bias,
                                                          want_dense=want_dense, spec_rows=spec_rows, info=info)
    return idx, val, _dense_or_empty(dense, x, W.shape[0]), recon, torch.tensor(info["flagged_rows"], dtype=torch.int64)


@_table_forward_prefilter.register_fake
def _(x, W, bias, Wq, meta, k, table, scale, dec_bias, want_dense, spec_rows):
    B, H = x.shape[0], W.shape[0]
    return _i32((B, k), x), _f32((B, k), x), _f32((B if want_dense else 0, H), x), _f32((B, x.shape[1]), x), _host_count()


@_op("densify")
def _densify(idx: Tensor, val: Tensor, H: int) -> Tensor:
    return _ops.densify(idx, val, H)


@_densify.register_fake
def _(idx, val, H):
    return _f32((idx.shape[0], H), idx)


# ---- BinarySAE dictionary ------------------------------------------------------------------------------------------
@_op("pack_binary")
def _pack_binary(logits: Tensor, D: int, n_bits: int) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (packed uint8 [H, row_bytes], polarize_sum float64 0-d, soft_gap float32 0-d)"""
    return _ops.pack_binary(logits, D, n_bits, want_polarize=True, want_soft_gap=True)


@_pack_binary.register_fake
def _(logits, D, n_bits):
    fw = 1 if n_bits <= 1 else 2 if n_bits <= 2 else 4 if n_bits <= 4 else 8
    row_bytes = ((D * fw + 31) // 32) * 4
    return (torch.empty((logits.shape[0], row_bytes), dtype=torch.uint8, device=logits.device),
            torch.empty((), dtype=torch.float64, device=logits.device), _f32((), logits))


@_op("unpack_binary")
def _unpack_binary(packed: Tensor, D: int, n_bits: int) -> Tensor:
    return _ops.unpack_binary(packed, D, n_bits)


@_unpack_binary.register_fake
def _(packed, D, n_bits):
    return _f32((packed.shape[0], D), packed)


@_op("binary_soft_table")
def _binary_soft_table(logits: Tensor, D: int, n_bits: int) -> Tensor:
    return _ops.binary_soft_table(logits, D, n_bits)


@_binary_soft_table.register_fake
def _(logits, D, n_bits):
    return _f32((logits.shape[0], D), logits)


@_op("decode_binary_sparse")
def _decode_binary_sparse(idx: Tensor, val: Tensor, packed: Tensor, D: int, n_bits: int, step: float,
                          bias: Optional[Tensor]) -> Tensor:
    return _ops.decode_binary_sparse(idx, val, packed, D, n_bits, step, bias)


@_decode_binary_sparse.register_fake
def _(idx, val, packed, D, n_bits, step, bias):
    return _f32((idx.shape[0], D), idx)


@_op("decode_table_sparse")
def _decode_table_sparse(idx: Tensor, val: Tensor, table: Tensor, scale: float, bias: Optional[Tensor]) -> Tensor:
    return _ops.decode_table_sparse(idx, val, table, scale, bias)


@_decode_table_sparse.register_fake
def _(idx, val, table, scale, bias):
    return _f32((idx.shape[0], table.shape[1]), idx)


# ---- ternary / matryoshka --------------------------------------------------------------------------------------------
@_op("pack_ternary")
def _pack_ternary(w: Tensor) -> Tensor:
    return _ops.pack_ternary(w)


@_pack_ternary.register_fake
def _(w):
    return _i32((w.shape[0], (w.shape[1] + 15) // 16), w)


@_op("decode_ternary_dense")
def _decode_ternary_dense(h: Tensor, codes: Tensor, D: int) -> Tensor:
    return _ops.decode_ternary_dense(h, codes, D)


@_decode_ternary_dense.register_fake
def _(h, codes, D):
    return _f32((h.shape[0], D), h)


@_op("pack_matryoshka")
def _pack_matryoshka(w: Tensor, wm: Tensor, n_bits: int, abs_range: float, sizes: Optional[List[int]]) -> Tuple[Tensor, Tensor]:
    return _ops.pack_matryoshka(w, wm, n_bits, abs_range, sizes)


@_pack_matryoshka.register_fake
def _(w, wm, n_bits, abs_range, sizes):
    return _i32((w.shape[1], (w.shape[0] + 15) // 16), w), _f32((w.shape[0],), w)


@_op("pack_matryoshka_rows")
def _pack_matryoshka_rows(w: Tensor, wm: Tensor) -> Tensor:
    return _ops.pack_matryoshka_rows(w, wm)


@_pack_matryoshka_rows.register_fake
def _(w, wm):
    return _i32((w.shape[0], (w.shape[1] + 7) // 8), w)


@_op("decode_matryoshka")
def _decode_matryoshka(zbits: Tensor, H: int, D: int, n_bits: int, codes: Tensor, scale: Tensor, bias: Optional[Tensor],
                       allow_bias: bool, sizes: Optional[List[int]], sparse: bool) -> Tuple[Tensor, Tensor]:
    """sparse: codes = the hidden-major dictionary (pack_matryoshka_rows), active units only; same outputs."""
    fn = _ops.decode_matryoshka_sparse if sparse else _ops.decode_matryoshka
    return fn(zbits, H, D, n_bits, codes, scale, bias, allow_bias, sizes)


@_decode_matryoshka.register_fake
def _(zbits, H, D, n_bits, codes, scale, bias, allow_bias, sizes, sparse):
    return _f32((n_bits, zbits.shape[0], D), zbits), torch.empty((n_bits,), dtype=torch.int64, device=zbits.device)


@_op("expand_codes_bf16")
def _expand_codes_bf16(codes: Tensor, D: int, H: int) -> Tensor:
    return _ops.expand_codes_bf16(codes, D, H)


@_expand_codes_bf16.register_fake
def _(codes, D, H):
    return torch.empty((H * D,), dtype=torch.bfloat16, device=codes.device)


@_op("decode_ternary_dense_split")
def _decode_ternary_dense_split(h: Tensor, tq: Tensor, D: int) -> Tensor:
    return _ops.decode_ternary_dense_split(h, tq, D)


@_decode_ternary_dense_split.register_fake
def _(h, tq, D):
    return _f32((h.shape[0], D), h)


@_op("split_scale_bf16")
def _split_scale_bf16(scale: Tensor) -> Tensor:
    return _ops.split_scale_bf16(scale)


@_split_scale_bf16.register_fake
def _(scale):
    return torch.empty((3, scale.shape[0]), dtype=torch.bfloat16, device=scale.device)


@_op("decode_matryoshka_split")
def _decode_matryoshka_split(zbits: Tensor, H: int, D: int, n_bits: int, tq: Tensor, s3: Tensor, bias: Optional[Tensor],
                             allow_bias: bool, sizes: Optional[List[int]]) -> Tuple[Tensor, Tensor]:
    return _ops.decode_matryoshka_split(zbits, H, D, n_bits, tq, s3, bias, allow_bias, sizes)


@_decode_matryoshka_split.register_fake
def _(zbits, H, D, n_bits, tq, s3, bias, allow_bias, sizes):
    return _f32((n_bits, zbits.shape[0], D), zbits), torch.empty((n_bits,), dtype=torch.int64, device=zbits.device)


@_op("pack_bits_gt")
def _pack_bits_gt(dense: Tensor, thr: float) -> Tensor:
    return _ops.pack_bits_gt(dense, thr)


@_pack_bits_gt.register_fake
def _(dense, thr):
    return _i32((dense.shape[0], (dense.shape[1] + 31) // 32), dense)


# ---- elementwise steps, metric, statistics ---------------------------------------------------------------------------
@_op("residual_update")
def _residual_update(residual: Tensor, recon: Tensor, scale: float) -> Tensor:
    return _ops.residual_update(residual, recon, scale)


@_residual_update.register_fake
def _(residual, recon, scale):
    return _f32(residual.shape, residual)


@_op("threshold_ge")
def _threshold_ge(pre: Tensor, cutoff: float) -> Tensor:
    return _ops.threshold_ge(pre, cutoff)


@_threshold_ge.register_fake
def _(pre, cutoff):
    return _f32(pre.shape, pre)


@_op("scale_bias_rows")
def _scale_bias_rows(acc: Tensor, scale: float, bias: Optional[Tensor]) -> Tensor:
    return _ops.scale_bias_rows(acc, scale, bias)


@_scale_bias_rows.register_fake
def _(acc, scale, bias):
    return _f32(acc.shape, acc)


@_op("sq_err_sum", mutates=("acc",))
def _sq_err_sum(recon: Tensor, x: Tensor, acc: Tensor) -> None:
    _ops.sq_err_sum(recon, x, acc)


@_op("activation_counts", mutates=("counts",))
def _activation_counts(idx: Tensor, val: Optional[Tensor], counts: Tensor) -> None:
    _ops.activation_counts(idx, val, counts.shape[0], counts)


@_op("activation_counts_bits", mutates=("counts",))
def _activation_counts_bits(zbits: Tensor, counts: Tensor) -> None:
    _ops.activation_counts_bits(zbits, counts)


@_op("coactivation_sparse", mutates=("coact",))
def _coactivation_sparse(idx: Tensor, val: Optional[Tensor], coact: Tensor) -> None:
    _ops.coactivation_sparse(idx, val, coact.shape[0], coact)


@_op("quantize_bits")
def _quantize_bits(x: Tensor, n_bits: int, scale_factor: float, signed: bool) -> Tensor:
    return _ops.quantize_bits(x, n_bits, scale_factor, signed)


@_quantize_bits.register_fake
def _(x, n_bits, scale_factor, signed):
    return _f32((x.shape[0], x.shape[1] * n_bits), x)


Q = torch.ops.qsae


# ---- the Python face the module classes call: ops.py's names and keyword defaults, routed through torch.ops.qsae -----------
def _none_if_empty(t: Tensor, want: bool):
    return t if want else None


def kperm_rows(src, out=None):
    return _ops.kperm_rows(src, out) if out is not None else Q.kperm_rows(src)


def encode_dense(x, W, bias, act=ACT_NONE, out=None, kperm=False):
    if out is not None or _ops.kernel_timer.enabled:                 # (out= variant and the bench's event brackets: not graph ops)
        return _ops.encode_dense(x, W, bias, act, out=out, kperm=kperm)
    return Q.encode_dense(x, W, bias, act, kperm)


def emu_pack_w(W):
    return Q.emu_pack_w(W)


def encode_dense_emu(x, Wc, meta2, bias, act=ACT_NONE):
    return Q.encode_dense_emu(x, Wc, meta2, bias, act)


def encode_bits(x, W, bias):
    return Q.encode_bits(x, W, bias)


def encode_bits_prefilter(x, W, bias, Wq, meta):
    z, flagged = Q.encode_bits_prefilter(x, W, bias, Wq, meta)
    return z, int(flagged)


def encode_bits_band(x, W, bias, Wq, meta):
    z, flagged = Q.encode_bits_band(x, W, bias, Wq, meta)
    return z, int(flagged)


def topk_rows(latent, k, zero_rest):
    return Q.topk_rows(latent, k, zero_rest)


def encode_topk(x, W, bias, k, kperm=False):
    return Q.encode_topk(x, W, bias, k, kperm)


def encode_topk_latent(x, W, bias, k, kperm=False):
    return Q.encode_topk_latent(x, W, bias, k, kperm)


def prefilter_pack_w(W, bias):
    return Q.prefilter_pack_w(W, bias)


def encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=True, dense_out=None, spec_rows=0, info=None):
    if dense_out is not None:
        return _ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense, dense_out, spec_rows, info)
    idx, val, dense, flagged = Q.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense, spec_rows)
    if info is not None:
        info["flagged_rows"] = int(flagged)
    return idx, val, _none_if_empty(dense, want_dense)


def binary_forward_prefilter(x, W, bias, Wq, meta, k, packed, n_bits, step, dec_bias, want_dense=True, spec_rows=0, info=None):
    idx, val, dense, recon, flagged = Q.binary_forward_prefilter(x, W, bias, Wq, meta, k, packed, n_bits, float(step), dec_bias,
                                                                 want_dense, spec_rows)
    if info is not None:
        info["flagged_rows"] = int(flagged)
    return idx, val, _none_if_empty(dense, want_dense), recon


def table_forward_prefilter(x, W, bias, Wq, meta, k, table, scale, dec_bias, want_dense=True, spec_rows=0, info=None):
    idx, val, dense, recon, flagged = Q.table_forward_prefilter(x, W, bias, Wq, meta, k, table, float(scale), dec_bias,
                                                                want_dense, spec_rows)
    if info is not None:
        info["flagged_rows"] = int(flagged)
    return idx, val, _none_if_empty(dense, want_dense), recon


def densify(idx, val, H, out=None):
    return _ops.densify(idx, val, H, out) if out is not None else Q.densify(idx, val, H)


def pack_binary(logits, D, n_bits, want_polarize=True, want_soft_gap=False):
    packed, pol, gap = Q.pack_binary(logits, D, n_bits)
    pol = pol if want_polarize else None
    return (packed, pol, gap) if want_soft_gap else (packed, pol)


def unpack_binary(packed, D, n_bits):
    return Q.unpack_binary(packed, D, n_bits)


def binary_soft_table(logits, D, n_bits):
    return Q.binary_soft_table(logits, D, n_bits)


def decode_binary_sparse(idx, val, packed, D, n_bits, step, bias=None):
    return Q.decode_binary_sparse(idx, val, packed, D, n_bits, float(step), bias)


def decode_table_sparse(idx, val, table, scale=1.0, bias=None):
    return Q.decode_table_sparse(idx, val, table, float(scale), bias)


def pack_ternary(w):
    return Q.pack_ternary(w)


def decode_ternary_dense(h, codes, D):
    return Q.decode_ternary_dense(h, codes, D)


def pack_matryoshka(w, wm, n_bits, abs_range, sizes=None):
    return Q.pack_matryoshka(w, wm, n_bits, float(abs_range), None if sizes is None else [int(s) for s in sizes])


def pack_matryoshka_rows(w, wm):
    return Q.pack_matryoshka_rows(w, wm)


def decode_matryoshka(zbits, H, D, n_bits, codes, scale, bias, allow_bias, sizes=None):
    return Q.decode_matryoshka(zbits, H, D, n_bits, codes, scale, bias, allow_bias, None if sizes is None else [int(s) for s in sizes],
                               False)


def decode_matryoshka_sparse(zbits, H, D, n_bits, codes_rows, scale, bias, allow_bias, sizes=None):
    return Q.decode_matryoshka(zbits, H, D, n_bits, codes_rows, scale, bias, allow_bias,
                               None if sizes is None else [int(s) for s in sizes], True)


def expand_codes_bf16(codes, D, H):
    return Q.expand_codes_bf16(codes, int(D), int(H))


def decode_ternary_dense_split(h, tq, D):
    return Q.decode_ternary_dense_split(h, tq, int(D))


def split_scale_bf16(scale):
    return Q.split_scale_bf16(scale)


def decode_matryoshka_split(zbits, H, D, n_bits, tq, s3, bias, allow_bias, sizes=None):
    return Q.decode_matryoshka_split(zbits, H, D, n_bits, tq, s3, bias, allow_bias, None if sizes is None else [int(v) for v in sizes])


def pack_bits_gt(dense, thr):
    return Q.pack_bits_gt(dense, float(thr))


def residual_update(residual, recon, scale=2.0):
    return Q.residual_update(residual, recon, float(scale))


def threshold_ge(pre, cutoff):
    return Q.threshold_ge(pre, float(cutoff))


def scale_bias_rows(acc, scale, bias):
    return Q.scale_bias_rows(acc, float(scale), bias)


def sq_err_sum(recon, x, acc=None):
    if acc is None:
        acc = torch.zeros((), dtype=torch.float64, device=x.device)
    Q.sq_err_sum(recon, x, acc)
    return acc


def activation_counts(idx, val, H, counts=None):
    if counts is None:
        counts = torch.zeros((H,), dtype=torch.int64, device=idx.device)
    Q.activation_counts(idx, val, counts)
    return counts


def activation_counts_bits(zbits, counts=None):
    if counts is None:
        counts = torch.zeros((32 * zbits.shape[1],), dtype=torch.int64, device=zbits.device)
    Q.activation_counts_bits(zbits, counts)
    return counts


def coactivation_sparse(idx, val, H, coact=None):
    if coact is None or coact.stride(0) != coact.shape[1]:
        return _ops.coactivation_sparse(idx, val, H, coact)
    Q.coactivation_sparse(idx, val, coact)
    return coact


def quantize_bits(x, n_bits, scale_factor, signed=True):
    return Q.quantize_bits(x, int(n_bits), float(scale_factor), bool(signed))


# what has no tensor result (shape queries, handles of batches in flight) stays plain Python
prefilter_supported = _ops.prefilter_supported
encode_bits_prefilter_supported = _ops.encode_bits_prefilter_supported
encode_bits_band_supported = _ops.encode_bits_band_supported
decode_matryoshka_sparse_supported = _ops.decode_matryoshka_sparse_supported
split_dec_supported = _ops.split_dec_supported
encode_dense_emu_supported = _ops.encode_dense_emu_supported
matryoshka_sizes = _ops.matryoshka_sizes
binary_row_bytes = _ops.binary_row_bytes
binary_forward_prefilter_submit = _ops.binary_forward_prefilter_submit
table_forward_prefilter_submit = _ops.table_forward_prefilter_submit
encode_bits_prefilter_submit = _ops.encode_bits_prefilter_submit
release_workspaces = _ops.release_workspaces
kernel_timer = _ops.kernel_timer


# ---- model level: one node per forward() ------------------------------------------------------------------------------------
_modules = weakref.WeakValueDictionary()
_next_handle = [1]


def module_handle(module) -> int:
    """Process-local integer under which a module's forward op finds the module (its derived-weight caches, its path
    switches).  Assigned once, in the constructor (GraphForwardMixin); a traced forward reads the plain attribute."""
    h = module.__dict__.get("_qsae_handle")
    if h is None or _modules.get(h) is not module:
        h = _next_handle[0]
        _next_handle[0] += 1
        module.__dict__["_qsae_handle"] = h
        _modules[h] = module
    return h


class GraphForwardMixin:
    """For the module classes: a handle of their own, also after ``copy.deepcopy`` and unpickling (a copied ``__dict__``
    would otherwise carry the original's handle, and the copy's graph node would run the original module)."""

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_qsae_handle":
                new.__dict__[k] = copy.deepcopy(v, memo)
        module_handle(new)
        return new

    def __setstate__(self, state):
        state = dict(state)
        state.pop("_qsae_handle", None)
        super().__setstate__(state)
        module_handle(self)


def _module(handle: int):
    m = _modules.get(handle)
    if m is None:
        raise RuntimeError(f"qsae: no live module behind handle {handle} (graphs that contain qsae::*_sae_forward nodes "
                           "are valid only in the process, and for the module, they were traced with)")
    return m


@_op("binary_sae_forward")
def _binary_sae_forward(x: Tensor, params: List[Tensor], handle: int, want_dense: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """BinarySAE.forward / forward_compact: -> (idx, val, dense latent ([0, H] when not wanted), reconstruction, polarize)"""
    m = _module(handle)
    idx, val, latent, recon = m._run(x, want_dense)
    # (the polarize loss is a cached per-checkpoint tensor: an op's outputs have to be fresh tensors, so it is copied)
    return idx, val, _dense_or_empty(latent, x, m.hidden_dim), recon, m.decoder.packed()["polarize"].clone()


@_binary_sae_forward.register_fake
def _(x, params, handle, want_dense):
    m = _module(handle)
    B, k = x.shape[0], m.top_k
    return (_i32((B, k), x), _f32((B, k), x), _f32((B if want_dense else 0, m.hidden_dim), x), _f32((B, m.input_dim), x),
            _f32((), x))


@_op("baseline_sae_forward")
def _baseline_sae_forward(x: Tensor, params: List[Tensor], handle: int, want_dense: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    m = _module(handle)
    idx, val, h, recon = m._run(x, want_dense)
    return idx, val, _dense_or_empty(h, x, m.encoder.linear.weight.shape[0]), recon


@_baseline_sae_forward.register_fake
def _(x, params, handle, want_dense):
    m = _module(handle)
    H, D = m.encoder.linear.weight.shape
    B, k = x.shape[0], m.topk
    return _i32((B, k), x), _f32((B, k), x), _f32((B if want_dense else 0, H), x), _f32((B, D), x)


@_op("ternary_sae_forward")
def _ternary_sae_forward(x: Tensor, params: List[Tensor], handle: int) -> Tuple[Tensor, Tensor]:
    return _module(handle)._forward_eager(x)


@_ternary_sae_forward.register_fake
def _(x, params, handle):
    m = _module(handle)
    H, D = m.encoder.linear.weight.shape
    return _f32((x.shape[0], H), x), _f32((x.shape[0], D), x)


@_op("levels_sae_forward")
def _levels_sae_forward(x: Tensor, params: List[Tensor], handle: int) -> Tuple[Tensor, Tensor]:
    """QuantizedMatryoshkaSAE / ResidualQuantizedSAE forward: -> (latent_groups [n], reconstruction_levels [n, B, D])"""
    groups, levels = _module(handle)._forward_eager(x)
    return torch.stack([g.reshape(()) for g in groups]), torch.stack(list(levels))


@_levels_sae_forward.register_fake
def _(x, params, handle):
    m = _module(handle)
    n = len(m.saes) if hasattr(m, "saes") else m.n_bits
    return _f32((n,), x), _f32((n, x.shape[0], m.input_dim), x)
