"""Tensor-level front-end of the C ABI (include/qsae.h): torch supplies device memory and the
current HIP stream, libqsae_hip.so does the work.  Every function requires CUDA (ROCm)
tensors and raises otherwise -- there is no CPU path in this package.
"""
from __future__ import annotations

import collections
import ctypes as C
import functools
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, check  # noqa: F401


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_tensor_device(fn):
    """Run `fn` with the device of its first tensor argument current: the C ABI (like any HIP library) launches on the
    current device, and a model moved to cuda:1 must not launch on cuda:0 because that is what the thread last used."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor):
                if a.is_cuda and a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


class _HipEventPair:
    """Two timing events for qsae_profile_sweep_events, created / read / destroyed through libqsae_hip.so itself
    (qsae_profile_event_*): the runtime that records them is the one that made them."""

    def __init__(self):
        lib = _lib.load()
        self._lib = lib
        self.a, self.b = C.c_void_p(), C.c_void_p()
        for e in (self.a, self.b):
            check(lib.qsae_profile_event_create(C.byref(e)))

    def elapsed_ms(self):
        ms = C.c_float(0.0)
        return ms.value if self._lib.qsae_profile_event_elapsed_ms(self.a, self.b, C.byref(ms)) == 0 else None

    def destroy(self):
        for e in (self.a, self.b):
            self._lib.qsae_profile_event_destroy(e)


class _KernelTimer:
    """Optional HIP-event brackets (used by bench.py to time the dominant kernel live inside the timed region).
    `encode_dense` launches are bracketed here; the candidate sweep is launched inside the library, which records a
    pair handed to it per call (sweep_pairs).  All events sit on the stream the kernel is launched on and are read only
    after the region has been synchronised."""

    def __init__(self):
        self.enabled = False
        self.sweep = False          # hand an event pair to every fused / prefilter call
        self._events = {}
        self._sweep_pairs = []

    def reset(self):
        self._events = {}
        for p in self._sweep_pairs:
            p.destroy()
        self._sweep_pairs = []

    def bracket(self, name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._events.setdefault(name, []).append((a, b))
        return a, b

    def mean_ms(self, name):
        ev = self._events.get(name)
        if not ev:
            return None
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def arm_sweep(self):
        """Called right before a library call that contains a candidate sweep."""
        if self.sweep:
            p = _HipEventPair()
            self._sweep_pairs.append(p)
            check(_lib.load().qsae_profile_sweep_events(p.a, p.b))

    def sweep_mean_ms(self):
        """-> (mean ms per sweep launch or None, launches); the region must have been synchronised."""
        ms = [m for m in (p.elapsed_ms() for p in self._sweep_pairs) if m is not None]
        return (sum(ms) / len(ms) if ms else None), len(ms)


kernel_timer = _KernelTimer()


def sweep_timing(enable: bool) -> None:
    """Time the candidate-sweep launch of every following fused / prefilter call (HIP events on the launch stream)."""
    kernel_timer.sweep = bool(enable)


def sweep_timing_collect(H: int):
    """-> (mean ms per sweep launch or None, launches, fraction of the encoder FLOPs per launch)."""
    ms, n = kernel_timer.sweep_mean_ms()
    return ms, n, float(_lib.load().qsae_profile_sweep_flop_fraction(H))


def _dev(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: quantizedsae_amd runs on MI355X only; tensor is on {t.device} "
                           "(no CPU fallback exists)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    _dev(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---- encoder ------------------------------------------------------------------------------
@_on_tensor_device
def kperm_rows(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K-interleaved copy of a [rows, K] fp32 matrix (K % 8 == 0); see qsae_kperm_rows."""
    src = _f32c(src, "src")
    rows, K = src.shape
    if out is None:
        out = torch.empty_like(src)
    check(_lib.load().qsae_kperm_rows(_p(src), rows, K, _p(out), _stream()))
    return out


@_on_tensor_device
def encode_dense(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE,
                 out: Optional[torch.Tensor] = None, kperm: bool = False) -> torch.Tensor:
    """kperm=True: x and W are already K-interleaved (kperm_rows)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    if W.shape[1] != D:
        raise ValueError(f"W is {tuple(W.shape)}, expected [H, {D}]")
    b = _f32c(bias, "bias") if bias is not None else None
    if out is None:
        out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    ev = kernel_timer.bracket("encode_dense") if kernel_timer.enabled else None
    if ev:
        ev[0].record()
    fn = _lib.load().qsae_encode_dense_kperm if kperm else _lib.load().qsae_encode_dense
    check(fn(_p(x), _p(W), _p(b), B, D, H, act, _p(out), out.stride(0) if B else H, _stream()))
    if ev:
        ev[1].record()
    return out


def encode_dense_emu_supported(D: int) -> bool:
    return D > 0 and D % 64 == 0


@_on_tensor_device
def emu_pack_w(W: torch.Tensor):
    """-> (Wc fp16 [H, 3 D], meta2 fp32 [2]) for encode_dense_emu (once per checkpoint)."""
    W = _f32c(W, "W")
    H, D = W.shape
    Wc = torch.empty((H, 3 * D), dtype=torch.float16, device=W.device)
    meta2 = torch.zeros((2,), dtype=torch.float32, device=W.device)
    check(_lib.load().qsae_emu_pack_w(_p(W), H, D, _p(Wc), _p(meta2), _stream()))
    return Wc, meta2


@_on_tensor_device
def encode_dense_emu(x: torch.Tensor, Wc: torch.Tensor, meta2: torch.Tensor, bias: Optional[torch.Tensor],
                     act: int = ACT_NONE) -> torch.Tensor:
    """encode_dense at fp32 accuracy (not bit-exactness) on the fp16 matrix pipe: see qsae_encode_dense_emu."""
    x = _f32c(x, "x")
    B, D = x.shape
    H = Wc.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    need = int(lib.qsae_encode_dense_emu_workspace_bytes(B, D)) if B > 0 else 1
    if need == 0:
        raise ValueError("shape not supported by the emulated encoder (D % 64 == 0)")
    ws = _workspace(x.device, need)
    check(lib.qsae_encode_dense_emu(_p(x), _p(Wc), _p(meta2), _p(b), B, D, H, act, _p(out), H, _p(ws), ws.numel(), _stream()))
    return out


@_on_tensor_device
def encode_bits(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """uint32-packed z bits [B, ceil(H/32)] (returned as int32 tensor)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    words = (H + 31) // 32
    z = torch.empty((B, words), dtype=torch.int32, device=x.device)
    check(_lib.load().qsae_encode_bits(_p(x), _p(W), _p(b), B, D, H, _p(z), words, _stream()))
    return z


def encode_bits_prefilter_supported(B: int, D: int, H: int) -> bool:
    return B > 0 and int(_lib.load().qsae_encode_bits_prefilter_workspace_bytes(B, D, H)) > 0


@_on_tensor_device
def encode_bits_prefilter(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                          meta: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """z bits identical to encode_bits, from the fp16 candidate sweep + exact re-evaluation of the latents near the
    cutoff.  Returns (zbits int32 [B, ceil(H/32)], rows that went through the exact dense kernel)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    if B == 0:
        return torch.empty((0, (H + 31) // 32), dtype=torch.int32, device=x.device), 0
    need = int(lib.qsae_encode_bits_prefilter_workspace_bytes(B, D, H))
    if need == 0:
        raise ValueError("shape not supported by the fp16 candidate sweep")
    ws = _workspace(x.device, need)
    words = (H + 31) // 32
    z = torch.empty((B, words), dtype=torch.int32, device=x.device)
    flagged = C.c_int(0)
    kernel_timer.arm_sweep()
    check(lib.qsae_encode_bits_prefilter(_p(x), _p(W), _p(b), _p(Wq), _p(meta), B, D, H, _p(z), words, _p(ws),
                                         ws.numel(), C.byref(flagged), _stream()))
    return z, int(flagged.value)


def encode_bits_band_supported(B: int, D: int, H: int) -> bool:
    return B > 0 and int(_lib.load().qsae_encode_bits_band_workspace_bytes(B, D, H)) > 0


@_on_tensor_device
def encode_bits_band(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                     meta: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """z bits identical to encode_bits for DENSE activations: every latent classified by an fp16 MFMA pass, the
    uncertainty band around the cutoff re-evaluated exactly.  Returns (zbits, rows that went through the exact kernel)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    if B == 0:
        return torch.empty((0, (H + 31) // 32), dtype=torch.int32, device=x.device), 0
    need = int(lib.qsae_encode_bits_band_workspace_bytes(B, D, H))
    if need == 0:
        raise ValueError("shape not supported by the fp16 band classification")
    ws = _workspace(x.device, need)
    words = (H + 31) // 32
    z = torch.empty((B, words), dtype=torch.int32, device=x.device)
    flagged = C.c_int(0)
    kernel_timer.arm_sweep()
    check(lib.qsae_encode_bits_band(_p(x), _p(W), _p(b), _p(Wq), _p(meta), B, D, H, _p(z), words, _p(ws), ws.numel(),
                                    C.byref(flagged), _stream()))
    return z, int(flagged.value)


@_on_tensor_device
def encode_bits_prefilter_submit(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                                 meta: torch.Tensor, slot: int = 0, band: bool = False, owner: int = 0) -> "PendingForward":
    """The two-call form of encode_bits_prefilter (qsae_encode_bits_prefilter_submit / _finish) or, with ``band``, of
    encode_bits_band: ``finish()`` returns the z bits; ``flagged_rows`` of the handle is the number of rows that went
    through the exact dense kernel."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    sizer = lib.qsae_encode_bits_band_workspace_bytes if band else lib.qsae_encode_bits_prefilter_workspace_bytes
    need = int(sizer(B, D, H)) if B > 0 else 1
    if need == 0:
        raise ValueError("shape not supported by the fp16 candidate sweep")
    slot = (owner, slot)
    _claim_slot(x.device, slot)
    ws = _workspace(x.device, need, slot, "pending")
    words = (H + 31) // 32
    z = torch.empty((B, words), dtype=torch.int32, device=x.device)
    cargs = (_p(x), _p(W), _p(b), _p(Wq), _p(meta), B, D, H, _p(z), words, _p(ws), ws.numel())
    if band:
        return _submit(lib.qsae_encode_bits_band_submit, lib.qsae_encode_bits_band_finish, cargs, (x, W, b, Wq, meta, ws), z,
                       x.device, slot)
    return _submit(lib.qsae_encode_bits_prefilter_submit, lib.qsae_encode_bits_prefilter_finish, cargs,
                   (x, W, b, Wq, meta, ws), z, x.device, slot)


@_on_tensor_device
def topk_rows(latent: torch.Tensor, k: int, zero_rest: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """In-place on `latent` when zero_rest.  Returns (idx int32 [B,k], val f32 [B,k])."""
    _dev(latent, "latent", torch.float32)
    if latent.dim() != 2 or latent.stride(1) != 1:
        raise ValueError("latent must be a 2-D tensor with unit inner stride")
    B, H = latent.shape
    idx = torch.empty((B, k), dtype=torch.int32, device=latent.device)
    val = torch.empty((B, k), dtype=torch.float32, device=latent.device)
    check(_lib.load().qsae_topk_rows(_p(latent), latent.stride(0) if B else H, B, H, k, _p(idx), _p(val),
                                     1 if zero_rest else 0, _stream()))
    return idx, val


_workspaces = collections.OrderedDict()     # (device, stream handle, kind, slot) -> uint8 tensor, least recently used first
#: scratch buffers kept per device (each is ~0.7 GB at the headline shape): the blocking calls of one stream need one,
#: every batch in flight another; streams / threads that have gone leave theirs behind until they fall off this list
WORKSPACE_CACHE_PER_DEVICE = 6
_busy_slots = {}                            # (device, stream handle, slot) -> PendingForward that owns the slot


def _workspace(device: torch.device, nbytes: int, slot: int = 0, kind: str = "call") -> torch.Tensor:
    """Scratch for one call in flight: one buffer per (device, stream, kind, slot).  Two streams never share one (their
    kernels would write the same candidate lists), and a buffer that has to grow is simply replaced: the old block goes
    back to the caching allocator, which reuses memory in the order of the stream it was allocated on -- the same
    stream every user of this buffer ran on.  kind "call" = the blocking entry points (whose use of the buffer ends with
    the call's last kernel), kind "pending" = submit / finish pairs, which own their buffer until finish() -- a blocking
    call between the two must not touch it -- with ``slot`` separating batches in flight together on one stream.
    The cache keeps the WORKSPACE_CACHE_PER_DEVICE most recently used buffers of a device (release_workspaces()
    drops all)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(device).cuda_stream, kind, slot)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    _workspaces.move_to_end(key)
    mine = [k for k in _workspaces if k[0] == dev]
    for k in mine[:max(0, len(mine) - WORKSPACE_CACHE_PER_DEVICE)]:
        ref = _busy_slots.get((k[0], k[1], k[3])) if k[2] == "pending" else None
        if ref is not None and ref() is not None:
            continue                        # a batch in flight still owns it
        del _workspaces[k]                  # (the block returns to the caching allocator, stream-ordered)
    return ws


def release_workspaces() -> None:
    """Drop every cached scratch buffer (they are re-created on demand).  Buffers of batches still in flight stay alive
    through their PendingForward handles."""
    _workspaces.clear()


@_on_tensor_device
def encode_topk(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], k: int, kperm: bool = False):
    """kperm=True: x and W are already K-interleaved (kperm_rows)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    need = int(lib.qsae_encode_topk_workspace_bytes(B, D, H, k))
    ws = _workspace(x.device, need)
    idx = torch.empty((B, k), dtype=torch.int32, device=x.device)
    val = torch.empty((B, k), dtype=torch.float32, device=x.device)
    fn = lib.qsae_encode_topk_kperm if kperm else lib.qsae_encode_topk
    kernel_timer.arm_sweep()
    check(fn(_p(x), _p(W), _p(b), B, D, H, k, _p(idx), _p(val), _p(ws), ws.numel(), _stream()))
    return idx, val


@_on_tensor_device
def encode_topk_latent(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], k: int, kperm: bool = False):
    """-> (idx, val, dense latent [B,H]); the dense tensor is zero-filled inside the encoder sweep."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    need = int(lib.qsae_encode_topk_workspace_bytes(B, D, H, k))
    ws = _workspace(x.device, need)
    idx = torch.empty((B, k), dtype=torch.int32, device=x.device)
    val = torch.empty((B, k), dtype=torch.float32, device=x.device)
    dense = torch.empty((B, H), dtype=torch.float32, device=x.device)
    kernel_timer.arm_sweep()
    check(lib.qsae_encode_topk_latent(_p(x), _p(W), _p(b), B, D, H, k, _p(idx), _p(val), _p(dense), H,
                                      1 if kperm else 0, _p(ws), ws.numel(), _stream()))
    return idx, val, dense


@_on_tensor_device
def prefilter_pack_w(W: torch.Tensor, bias: Optional[torch.Tensor]):
    """-> (Wq fp16 [H, D], meta fp32 [4]) for encode_topk_prefilter (once per checkpoint)."""
    W = _f32c(W, "W")
    H, D = W.shape
    b = _f32c(bias, "bias") if bias is not None else None
    Wq = torch.empty((H, D), dtype=torch.float16, device=W.device)
    meta = torch.zeros((4,), dtype=torch.float32, device=W.device)
    check(_lib.load().qsae_prefilter_pack_w(_p(W), _p(b), H, D, _p(Wq), _p(meta), _stream()))
    return Wq, meta


def prefilter_supported(B: int, D: int, H: int, k: int) -> bool:
    return int(_lib.load().qsae_encode_topk_prefilter_workspace_bytes(B, D, H, k)) > 0


@_on_tensor_device
def encode_topk_prefilter(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                          meta: torch.Tensor, k: int, want_dense: bool = True, dense_out: Optional[torch.Tensor] = None,
                          spec_rows: int = 0, info: Optional[dict] = None):
    """fp16-prefiltered encoder + exact top-k (+ dense latent): results identical to encode_topk_latent.
    ``dense_out``: optional [B, >=H] fp32 buffer (row stride a multiple of 4) that receives the dense latent.
    ``spec_rows``: flagged rows the device recomputes while the host waits for their count (see qsae.h).
    ``info``: a dict that receives ``flagged_rows`` (rows that went through the exact fallback kernels)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    lib = _lib.load()
    need = int(lib.qsae_encode_topk_prefilter_workspace_bytes(B, D, H, k))
    if need == 0:
        raise ValueError("shape not supported by the fp16 prefilter")
    ws = _workspace(x.device, need)
    idx = torch.empty((B, k), dtype=torch.int32, device=x.device)
    val = torch.empty((B, k), dtype=torch.float32, device=x.device)
    if dense_out is not None:
        _dev(dense_out, "dense_out", torch.float32)
        if dense_out.dim() != 2 or dense_out.shape[0] != B or dense_out.shape[1] < H or dense_out.stride(1) != 1:
            raise ValueError("dense_out must be a [B, >=H] fp32 tensor with unit column stride")
        dense, ld = dense_out, dense_out.stride(0)
    else:
        dense = torch.empty((B, H), dtype=torch.float32, device=x.device) if want_dense else None
        ld = H
    flagged = C.c_int(0)
    kernel_timer.arm_sweep()
    check(lib.qsae_encode_topk_prefilter(_p(x), _p(W), _p(b), _p(Wq), _p(meta), B, D, H, k, _p(idx), _p(val),
                                         _p(dense), ld, _p(ws), ws.numel(), int(spec_rows), C.byref(flagged), _stream()))
    if info is not None:
        info["flagged_rows"] = int(flagged.value)
    return idx, val, (dense[:, :H] if dense_out is not None else dense)


def _decode_prefilter_args(x, W, bias, Wq, meta, k, decoder, dec_bias, want_dense, slot, kind):
    """Argument tuple shared by the forward-in-one-call entry points.  ``decoder`` = ("packed", packed uint8 [H, row_bytes],
    n_bits, step) or ("table", fp32 [H, D], scale)."""
    x, W = _f32c(x, "x"), _f32c(W, "W")
    B, D = x.shape
    H = W.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    db = _f32c(dec_bias, "dec_bias") if dec_bias is not None else None
    if decoder[0] == "packed":
        dict_t = _dev(decoder[1], "packed", torch.uint8)
        dargs = (_p(dict_t), int(decoder[2]), float(decoder[3]))
    else:
        dict_t = _f32c(decoder[1], "table")
        if tuple(dict_t.shape) != (H, D):
            raise ValueError(f"table is {tuple(dict_t.shape)}, expected [{H}, {D}]")
        dargs = (_p(dict_t), float(decoder[2]))
    need = int(_lib.load().qsae_encode_topk_prefilter_workspace_bytes(B, D, H, k))
    if need == 0:
        raise ValueError("shape not supported by the fp16 prefilter")
    ws = _workspace(x.device, need, slot, kind)
    idx = torch.empty((B, k), dtype=torch.int32, device=x.device)
    val = torch.empty((B, k), dtype=torch.float32, device=x.device)
    dense = torch.empty((B, H), dtype=torch.float32, device=x.device) if want_dense else None
    recon = torch.empty((B, D), dtype=torch.float32, device=x.device)
    cargs = (_p(x), _p(W), _p(b), _p(Wq), _p(meta), B, D, H, k) + dargs + (_p(db), _p(idx), _p(val), _p(dense), H, _p(recon),
                                                                          _p(ws), ws.numel())
    keep = (x, W, b, Wq, meta, dict_t, db, ws)          # referenced by the pointers above
    return cargs, keep, (idx, val, dense, recon)


@_on_tensor_device
def binary_forward_prefilter(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                             meta: torch.Tensor, k: int, packed: torch.Tensor, n_bits: int, step: float,
                             dec_bias: Optional[torch.Tensor], want_dense: bool = True, spec_rows: int = 0,
                             info: Optional[dict] = None):
    """encode_topk_prefilter + decode_binary_sparse in one call (rows are decoded by the refinement kernel as it
    ranks them): (idx, val, dense latent or None, reconstruction), bit-identical to the two separate calls."""
    cargs, keep, outs = _decode_prefilter_args(x, W, bias, Wq, meta, k, ("packed", packed, n_bits, step), dec_bias,
                                               want_dense, 0, "call")
    flagged = C.c_int(0)
    kernel_timer.arm_sweep()
    check(_lib.load().qsae_binary_forward_prefilter(*cargs, int(spec_rows), C.byref(flagged), _stream()))
    if info is not None:
        info["flagged_rows"] = int(flagged.value)
    return outs


@_on_tensor_device
def table_forward_prefilter(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                            meta: torch.Tensor, k: int, table: torch.Tensor, scale: float,
                            dec_bias: Optional[torch.Tensor], want_dense: bool = True, spec_rows: int = 0,
                            info: Optional[dict] = None):
    """encode_topk_prefilter + decode_table_sparse in one call (fp32 dictionary rows [H, D]: the Baseline decoder, the
    soft integers of an unpolarised BinarySAE): (idx, val, dense latent or None, reconstruction), bit-identical to the
    two separate calls."""
    cargs, keep, outs = _decode_prefilter_args(x, W, bias, Wq, meta, k, ("table", table, scale), dec_bias, want_dense, 0,
                                               "call")
    flagged = C.c_int(0)
    kernel_timer.arm_sweep()
    check(_lib.load().qsae_table_forward_prefilter(*cargs, int(spec_rows), C.byref(flagged), _stream()))
    if info is not None:
        info["flagged_rows"] = int(flagged.value)
    return outs


class PendingForward:
    """A batch whose main kernels are queued (*_submit).  ``finish()`` waits for the 4-byte count of rows that need
    the exact fallback -- by then the GPU is usually busy with the NEXT batch's kernels -- enqueues that fallback ON THE
    STREAM THE BATCH WAS SUBMITTED ON and returns the outputs, which must not be read before ``finish()`` has returned.
    The handle owns its workspace slot from submit to finish: a second submit into the same slot raises, and the
    blocking entry points use buffers of their own, so nothing between the two calls can disturb the lists the fallback
    reads.  Not safe to share between threads."""

    def __init__(self, finish_fn, cargs, keep, outs, word, event, device, stream, slot_key):
        self._finish_fn, self._cargs, self._keep, self._outs = finish_fn, cargs, keep, outs
        self._word, self._event, self._device, self._stream, self._slot_key = word, event, device, stream, slot_key
        self.flagged_rows = None
        _busy_slots[slot_key] = weakref.ref(self)       # (weak: a dropped handle must still be collected)

    def _release(self):
        ref = _busy_slots.get(self._slot_key)
        if ref is not None and ref() in (self, None):
            del _busy_slots[self._slot_key]
        self._cargs = self._keep = None

    def finish(self):
        if self._cargs is None:
            return self._outs
        self._event.synchronize()                      # the count has landed in the pinned word
        self.flagged_rows = int(self._word.item())
        try:
            with torch.cuda.device(self._device), torch.cuda.stream(self._stream):
                check(self._finish_fn(*self._cargs, self.flagged_rows, _stream()))
        finally:
            self._release()
        return self._outs

    def __del__(self):
        # dropped without finish(): the asynchronous 4-byte copy may still be pending -- the pinned word must outlive it
        try:
            if self._cargs is not None:
                self._event.synchronize()
                self._release()
        except Exception:
            pass


def _submit(submit_fn, finish_fn, cargs, keep, outs, device, slot):
    stream = torch.cuda.current_stream(device)
    slot_key = (device.index if device.index is not None else torch.cuda.current_device(), stream.cuda_stream, slot)
    word = torch.zeros((1,), dtype=torch.int32).pin_memory()
    kernel_timer.arm_sweep()
    check(submit_fn(*cargs, C.c_void_p(word.data_ptr()), _stream()))
    ev = torch.cuda.Event()
    ev.record(stream)
    return PendingForward(finish_fn, cargs, keep, outs, word, ev, device, stream, slot_key)


def _claim_slot(device, slot):
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream, slot)
    ref = _busy_slots.get(key)
    if ref is not None and ref() is None:
        del _busy_slots[key]                # its handle is gone (collected without finish)
        ref = None
    if ref is not None:
        raise RuntimeError(f"submit: slot {slot[-1] if isinstance(slot, tuple) else slot} of this stream still holds a batch whose finish() / result() has not been "
                           "called; batches in flight together need different slot numbers")


@_on_tensor_device
def binary_forward_prefilter_submit(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                                    meta: torch.Tensor, k: int, packed: torch.Tensor, n_bits: int, step: float,
                                    dec_bias: Optional[torch.Tensor], want_dense: bool = True,
                                    slot: int = 0, owner: int = 0) -> PendingForward:
    """The two-call form of binary_forward_prefilter (qsae_prefilter_submit / _finish): nothing in here waits for the
    GPU, so the caller can submit batch i+1 before it finishes batch i.  Batches in flight together on one stream
    need different ``slot`` numbers (each slot is a workspace of its own); finish them in submission order."""
    slot = (owner, slot)            # slots are per owner (module): two models on one stream do not share workspaces
    _claim_slot(x.device, slot)
    cargs, keep, outs = _decode_prefilter_args(x, W, bias, Wq, meta, k, ("packed", packed, n_bits, step), dec_bias,
                                               want_dense, slot, "pending")
    lib = _lib.load()
    return _submit(lib.qsae_prefilter_submit, lib.qsae_prefilter_finish, cargs, keep, outs, x.device, slot)


@_on_tensor_device
def table_forward_prefilter_submit(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], Wq: torch.Tensor,
                                   meta: torch.Tensor, k: int, table: torch.Tensor, scale: float,
                                   dec_bias: Optional[torch.Tensor], want_dense: bool = True,
                                   slot: int = 0, owner: int = 0) -> PendingForward:
    """The two-call form of table_forward_prefilter (qsae_prefilter_submit_table / _finish_table)."""
    slot = (owner, slot)
    _claim_slot(x.device, slot)
    cargs, keep, outs = _decode_prefilter_args(x, W, bias, Wq, meta, k, ("table", table, scale), dec_bias, want_dense, slot,
                                               "pending")
    lib = _lib.load()
    return _submit(lib.qsae_prefilter_submit_table, lib.qsae_prefilter_finish_table, cargs, keep, outs, x.device, slot)


@_on_tensor_device
def densify(idx: torch.Tensor, val: torch.Tensor, H: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(idx, "idx", torch.int32)
    _dev(val, "val", torch.float32)
    B, k = idx.shape
    if out is None:
        out = torch.empty((B, H), dtype=torch.float32, device=idx.device)
    check(_lib.load().qsae_densify(_p(idx.contiguous()), _p(val.contiguous()), B, k, H, _p(out),
                                   out.stride(0) if B else H, _stream()))
    return out


# ---- BinarySAE dictionary -----------------------------------------------------------------
def binary_row_bytes(D: int, n_bits: int) -> int:
    r = int(_lib.load().qsae_binary_row_bytes(D, n_bits))
    if r < 0:
        check(r)
    return r


@_on_tensor_device
def pack_binary(logits: torch.Tensor, D: int, n_bits: int, want_polarize: bool = True, want_soft_gap: bool = False):
    """-> (packed uint8 [H, row_bytes], polarize_sum float64 0-d tensor or None[, soft_gap float32 0-d tensor]).
    soft_gap = max |soft integer - hard integer| over the dictionary (see qsae_pack_binary)."""
    logits = _f32c(logits, "logits")
    H = logits.shape[0]
    if logits.shape[1] != D * n_bits:
        raise ValueError(f"logits is {tuple(logits.shape)}, expected [H, {D * n_bits}]")
    packed = torch.empty((H, binary_row_bytes(D, n_bits)), dtype=torch.uint8, device=logits.device)
    pol = torch.zeros((), dtype=torch.float64, device=logits.device) if want_polarize else None
    gap = torch.zeros((), dtype=torch.float32, device=logits.device) if want_soft_gap else None
    check(_lib.load().qsae_pack_binary(_p(logits), H, D, n_bits, _p(packed), _p(pol), _p(gap), _stream()))
    return (packed, pol, gap) if want_soft_gap else (packed, pol)


@_on_tensor_device
def unpack_binary(packed: torch.Tensor, D: int, n_bits: int) -> torch.Tensor:
    _dev(packed, "packed", torch.uint8)
    H = packed.shape[0]
    out = torch.empty((H, D), dtype=torch.float32, device=packed.device)
    check(_lib.load().qsae_unpack_binary(_p(packed), H, D, n_bits, _p(out), _stream()))
    return out


@_on_tensor_device
def binary_soft_table(logits: torch.Tensor, D: int, n_bits: int) -> torch.Tensor:
    logits = _f32c(logits, "logits")
    H = logits.shape[0]
    out = torch.empty((H, D), dtype=torch.float32, device=logits.device)
    check(_lib.load().qsae_binary_soft_table(_p(logits), H, D, n_bits, _p(out), _stream()))
    return out


@_on_tensor_device
def decode_binary_sparse(idx, val, packed, D: int, n_bits: int, step: float, bias=None) -> torch.Tensor:
    _dev(idx, "idx", torch.int32)
    _dev(val, "val", torch.float32)
    _dev(packed, "packed", torch.uint8)
    B, k = idx.shape
    H = packed.shape[0]
    b = _f32c(bias, "bias") if bias is not None else None
    recon = torch.empty((B, D), dtype=torch.float32, device=idx.device)
    check(_lib.load().qsae_decode_binary_sparse(_p(idx.contiguous()), _p(val.contiguous()), B, k, _p(packed), H, D,
                                                n_bits, float(step), _p(b), _p(recon), _stream()))
    return recon


@_on_tensor_device
def decode_table_sparse(idx, val, table: torch.Tensor, scale: float = 1.0, bias=None) -> torch.Tensor:
    _dev(idx, "idx", torch.int32)
    _dev(val, "val", torch.float32)
    table = _f32c(table, "table")
    B, k = idx.shape
    H, D = table.shape
    b = _f32c(bias, "bias") if bias is not None else None
    recon = torch.empty((B, D), dtype=torch.float32, device=idx.device)
    check(_lib.load().qsae_decode_table_sparse(_p(idx.contiguous()), _p(val.contiguous()), B, k, _p(table), H, D,
                                               float(scale), _p(b), _p(recon), _stream()))
    return recon


# ---- ternary ---------------------------------------------------------------------------------
@_on_tensor_device
def pack_ternary(w: torch.Tensor) -> torch.Tensor:
    """decoder.weight [D, H] -> 2-bit codes int32 [D, ceil(H/16)]."""
    w = _f32c(w, "w")
    D, H = w.shape
    codes = torch.empty((D, (H + 15) // 16), dtype=torch.int32, device=w.device)
    check(_lib.load().qsae_pack_ternary(_p(w), D, H, _p(codes), _stream()))
    return codes


@_on_tensor_device
def decode_ternary_dense(h: torch.Tensor, codes: torch.Tensor, D: int) -> torch.Tensor:
    _dev(h, "h", torch.float32)
    B, H = h.shape
    recon = torch.empty((B, D), dtype=torch.float32, device=h.device)
    check(_lib.load().qsae_decode_ternary_dense(_p(h), h.stride(0) if B else H, B, H, _p(codes), D, _p(recon),
                                                _stream()))
    return recon


# ---- dense decoders on the bf16 matrix pipe ----------------------------------------------------
def split_dec_supported(B: int, H: int, D: int) -> bool:
    return bool(_lib.load().qsae_split_dec_supported(int(B), int(H), int(D)))


@_on_tensor_device
def expand_codes_bf16(codes: torch.Tensor, D: int, H: int) -> torch.Tensor:
    """2-bit codes [D, ceil(H/16)] (pack_ternary / pack_matryoshka) -> the bf16 dictionary image the split decoders
    stream (opaque, 2 H D bytes; once per checkpoint)."""
    _dev(codes, "codes", torch.int32)
    lib = _lib.load()
    nbytes = int(lib.qsae_expand_codes_bf16_bytes(D, H))
    if nbytes == 0:
        raise ValueError("shape not supported by the bf16 split decoders (D == 512, H % 64 == 0)")
    tq = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=codes.device)
    check(lib.qsae_expand_codes_bf16(_p(codes.contiguous()), D, H, _p(tq), _stream()))
    return tq


@_on_tensor_device
def decode_ternary_dense_split(h: torch.Tensor, tq: torch.Tensor, D: int) -> torch.Tensor:
    """decode_ternary_dense from three exact bf16 terms of h on v_mfma_f32_32x32x16_bf16 (fp32 accumulation)."""
    _dev(h, "h", torch.float32)
    B, H = h.shape
    recon = torch.empty((B, D), dtype=torch.float32, device=h.device)
    check(_lib.load().qsae_decode_ternary_dense_split(_p(h), h.stride(0) if B else H, B, H, _p(tq), D, _p(recon), _stream()))
    return recon


@_on_tensor_device
def split_scale_bf16(scale: torch.Tensor) -> torch.Tensor:
    """scale [H] (pack_matryoshka) -> the three bf16 terms of 2 * scale, [3, H] bfloat16."""
    scale = _f32c(scale, "scale")
    H = scale.shape[0]
    s3 = torch.empty((3, H), dtype=torch.bfloat16, device=scale.device)
    check(_lib.load().qsae_split_scale_bf16(_p(scale), H, _p(s3), _stream()))
    return s3


@_on_tensor_device
def decode_matryoshka_split(zbits: torch.Tensor, H: int, D: int, n_bits: int, tq, s3, bias, allow_bias: bool, sizes=None):
    """decode_matryoshka on the bf16 matrix pipe: -> (levels f32 [n_bits, B, D], l0_counts int64 [n_bits])."""
    _dev(zbits, "zbits", torch.int32)
    B = zbits.shape[0]
    levels = torch.empty((n_bits, B, D), dtype=torch.float32, device=zbits.device)
    counts = torch.zeros((n_bits,), dtype=torch.int64, device=zbits.device)
    b = _f32c(bias, "bias") if bias is not None else None
    keep, sp = _sizes_arg(sizes, n_bits)
    check(_lib.load().qsae_decode_matryoshka_split(_p(zbits), zbits.stride(0) if B else (H + 31) // 32, B, H, D, n_bits, sp,
                                                   _p(tq), _p(s3), _p(b), 1 if allow_bias else 0, _p(levels), _p(counts),
                                                   _stream()))
    return levels, counts


# ---- matryoshka ------------------------------------------------------------------------------
def matryoshka_sizes(H: int, n_bits: int):
    arr = (C.c_int32 * n_bits)()
    check(_lib.load().qsae_matryoshka_sizes(H, n_bits, C.cast(arr, C.c_void_p)))
    return [int(v) for v in arr]


def _sizes_arg(sizes, n_bits):
    if sizes is None:
        return None, C.c_void_p(0)
    if len(sizes) != n_bits:
        raise ValueError("level sizes must have n_bits entries")
    arr = (C.c_int32 * n_bits)(*[int(s) for s in sizes])
    return arr, C.cast(arr, C.c_void_p)


@_on_tensor_device
def pack_matryoshka(w: torch.Tensor, wm: torch.Tensor, n_bits: int, abs_range: float, sizes=None):
    """-> (codes int32 [D, ceil(H/16)] of S/2, scale fp32 [H])."""
    w, wm = _f32c(w, "w"), _f32c(wm, "wm")
    H, D = w.shape
    codes = torch.empty((D, (H + 15) // 16), dtype=torch.int32, device=w.device)
    scale = torch.empty((H,), dtype=torch.float32, device=w.device)
    keep, sp = _sizes_arg(sizes, n_bits)
    check(_lib.load().qsae_pack_matryoshka(_p(w), _p(wm), H, D, n_bits, float(abs_range), sp, _p(codes), _p(scale),
                                           _stream()))
    return codes, scale


@_on_tensor_device
def decode_matryoshka(zbits: torch.Tensor, H: int, D: int, n_bits: int, codes, scale, bias, allow_bias: bool,
                      sizes=None):
    """-> (levels f32 [n_bits, B, D], l0_counts int64 [n_bits])."""
    _dev(zbits, "zbits", torch.int32)
    B = zbits.shape[0]
    levels = torch.empty((n_bits, B, D), dtype=torch.float32, device=zbits.device)
    counts = torch.zeros((n_bits,), dtype=torch.int64, device=zbits.device)
    b = _f32c(bias, "bias") if bias is not None else None
    keep, sp = _sizes_arg(sizes, n_bits)
    check(_lib.load().qsae_decode_matryoshka(_p(zbits), zbits.stride(0) if B else (H + 31) // 32, B, H, D, n_bits,
                                             sp, _p(codes), _p(scale), _p(b), 1 if allow_bias else 0, _p(levels),
                                             _p(counts), _stream()))
    return levels, counts


@_on_tensor_device
def pack_matryoshka_rows(w: torch.Tensor, wm: torch.Tensor) -> torch.Tensor:
    """-> codes_rows int32 [H, ceil(D/8)]: the dictionary in hidden-major order (4-bit fields) for decode_matryoshka_sparse."""
    w, wm = _f32c(w, "w"), _f32c(wm, "wm")
    H, D = w.shape
    codes = torch.empty((H, (D + 7) // 8), dtype=torch.int32, device=w.device)
    check(_lib.load().qsae_pack_matryoshka_rows(_p(w), _p(wm), H, D, _p(codes), _stream()))
    return codes


def decode_matryoshka_sparse_supported(D: int) -> bool:
    return D in (64, 128, 256, 512, 1024)


@_on_tensor_device
def decode_matryoshka_sparse(zbits: torch.Tensor, H: int, D: int, n_bits: int, codes_rows, scale, bias,
                             allow_bias: bool, sizes=None):
    """decode_matryoshka on the active units only (same outputs)."""
    _dev(zbits, "zbits", torch.int32)
    B = zbits.shape[0]
    levels = torch.empty((n_bits, B, D), dtype=torch.float32, device=zbits.device)
    counts = torch.zeros((n_bits,), dtype=torch.int64, device=zbits.device)
    b = _f32c(bias, "bias") if bias is not None else None
    keep, sp = _sizes_arg(sizes, n_bits)
    check(_lib.load().qsae_decode_matryoshka_sparse(_p(zbits), zbits.stride(0) if B else (H + 31) // 32, B, H, D,
                                                    n_bits, sp, _p(codes_rows), _p(scale), _p(b),
                                                    1 if allow_bias else 0, _p(levels), _p(counts), _stream()))
    return levels, counts


@_on_tensor_device
def pack_bits_gt(dense: torch.Tensor, thr: float) -> torch.Tensor:
    """int32-packed bits [B, ceil(H/32)] of (dense > thr)."""
    _dev(dense, "dense", torch.float32)
    if dense.stride(1) != 1:
        dense = dense.contiguous()
    B, H = dense.shape
    words = (H + 31) // 32
    z = torch.empty((B, words), dtype=torch.int32, device=dense.device)
    check(_lib.load().qsae_pack_bits_gt(_p(dense), dense.stride(0) if B else H, B, H, float(thr), _p(z), words,
                                        _stream()))
    return z


# ---- small elementwise steps ---------------------------------------------------------------------
@_on_tensor_device
def residual_update(residual: torch.Tensor, recon: torch.Tensor, scale: float = 2.0) -> torch.Tensor:
    """(residual - recon) * scale, each operation rounded separately (sae/residual_quantized.py:67)."""
    residual, recon = _f32c(residual, "residual"), _f32c(recon, "recon")
    if residual.shape != recon.shape:
        raise ValueError("residual and recon must have the same shape")
    out = torch.empty_like(residual)
    check(_lib.load().qsae_residual_update(_p(residual), _p(recon), residual.numel(), float(scale), _p(out), _stream()))
    return out


@_on_tensor_device
def threshold_ge(pre: torch.Tensor, cutoff: float) -> torch.Tensor:
    """1.0 where pre >= cutoff else 0.0 (sae/binary_latent.py:21-24 with the fp32 cutoff of sigmoid >= 0.5)."""
    pre = _f32c(pre, "pre")
    out = torch.empty_like(pre)
    check(_lib.load().qsae_threshold_ge(_p(pre), pre.numel(), float(cutoff), _p(out), _stream()))
    return out


@_on_tensor_device
def scale_bias_rows(acc: torch.Tensor, scale: float, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """scale * acc + bias over the rows of a [B, D] tensor, multiply and add rounded separately (sae/binary.py:38)."""
    acc = _f32c(acc, "acc")
    B, D = acc.shape
    b = _f32c(bias, "bias") if bias is not None else None
    out = torch.empty_like(acc)
    check(_lib.load().qsae_scale_bias_rows(_p(acc), B, D, float(scale), _p(b), _p(out), _stream()))
    return out


# ---- metric ------------------------------------------------------------------------------------
@_on_tensor_device
def sq_err_sum(recon: torch.Tensor, x: torch.Tensor, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """acc (float64 0-d device tensor) += sum((recon - x)^2); returns acc (no host sync)."""
    recon, x = _f32c(recon, "recon"), _f32c(x, "x")
    if recon.shape != x.shape:
        raise ValueError("recon and x must have the same shape")
    if acc is None:
        acc = torch.zeros((), dtype=torch.float64, device=x.device)
    check(_lib.load().qsae_sq_err_sum(_p(recon), _p(x), recon.numel(), _p(acc), _stream()))
    return acc


# ---- consumers of the sparse latent (activation statistics) ---------------------------------------------
@_on_tensor_device
def activation_counts(idx: torch.Tensor, val: Optional[torch.Tensor], H: int,
                      counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """counts[h] += #rows whose entry h is active (val > 0; every listed entry when val is None).  int64 [H]."""
    _dev(idx, "idx", torch.int32)
    B, k = idx.shape
    if val is not None:
        _dev(val, "val", torch.float32)
    if counts is None:
        counts = torch.zeros((H,), dtype=torch.int64, device=idx.device)
    check(_lib.load().qsae_activation_counts(_p(idx.contiguous()), _p(val.contiguous()) if val is not None else None,
                                             B, k, H, _p(counts), _stream()))
    return counts


@_on_tensor_device
def activation_counts_bits(zbits: torch.Tensor, counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """counts[32 w + j] += popcount over rows of bit j of word w.  zbits int32 [B, words]; int64 [32 * words]."""
    _dev(zbits, "zbits", torch.int32)
    B, words = zbits.shape
    if counts is None:
        counts = torch.zeros((32 * words,), dtype=torch.int64, device=zbits.device)
    check(_lib.load().qsae_activation_counts_bits(_p(zbits), zbits.stride(0) if B else words, B, 32 * words, _p(counts),
                                                  _stream()))
    return counts


@_on_tensor_device
def coactivation_sparse(idx: torch.Tensor, val: Optional[torch.Tensor], H: int,
                        coact: Optional[torch.Tensor] = None) -> torch.Tensor:
    """coact[a, c] += #rows in which units a and c are both active (mask^T @ mask).  int32 [H, H]."""
    _dev(idx, "idx", torch.int32)
    B, k = idx.shape
    if val is not None:
        _dev(val, "val", torch.float32)
    if coact is None:
        coact = torch.zeros((H, H), dtype=torch.int32, device=idx.device)
    check(_lib.load().qsae_coactivation_sparse(_p(idx.contiguous()), _p(val.contiguous()) if val is not None else None,
                                               B, k, H, _p(coact), coact.stride(0), _stream()))
    return coact


@_on_tensor_device
def quantize_bits(x: torch.Tensor, n_bits: int, scale_factor: float, signed: bool = True) -> torch.Tensor:
    """n-bit code of every activation as LSB-first 0/1 floats, [B, D * n_bits] (data/dataset.py:76-102)."""
    x = _f32c(x, "x")
    B, D = x.shape
    out = torch.empty((B, D * n_bits), dtype=torch.float32, device=x.device)
    check(_lib.load().qsae_quantize_bits(_p(x), D, B, D, int(n_bits), float(scale_factor), 1 if signed else 0, _p(out),
                                         _stream()))
    return out
