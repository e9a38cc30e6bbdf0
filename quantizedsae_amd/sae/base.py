"""Base class and shared plumbing (reference: sae/base.py:5-29)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)


class HipEncoder(nn.Sequential):
    """``nn.Sequential(nn.Linear[, activation])`` whose forward is one HIP kernel.

    Keeping the Sequential container preserves the reference's state_dict keys
    (``encoder.0.weight`` / ``encoder.0.bias``) and lets callers keep invoking
    ``model.encoder(x)`` (scripts/analysis/dynamic_analysis.py:51); the contraction, bias and
    activation run in ``qsae_encode_dense`` instead of ATen.
    """

    def __init__(self, linear: nn.Linear, activation: Optional[nn.Module] = None):
        mods = [linear] + ([activation] if activation is not None else [])
        super().__init__(*mods)
        if activation is None:
            self._act = ops.ACT_NONE
        elif isinstance(activation, nn.ReLU):
            self._act = ops.ACT_RELU
        elif isinstance(activation, nn.Sigmoid):
            self._act = ops.ACT_SIGMOID
        else:
            raise TypeError(f"unsupported encoder activation {type(activation)}")

    @property
    def linear(self) -> nn.Linear:
        return self[0]

    #: "fp32" (default): the exact fmaf chain of the fp32 matrix pipe, bit-identical to the oracle.  "emulated": fp32 ACCURACY
    #: on the fp16 matrix pipe (both operands as two fp16 terms, three partial contractions, every product exact, fp32
    #: accumulation; qsae_encode_dense_emu) -- differs from the chain by accumulation-order noise (~1e-6 of a latent's standard
    #: deviation), about twice as fast.  Only this dense call uses it; the top-k / threshold paths always rank exact latents.
    precision = "fp32"

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        with torch.no_grad():
            if self.precision not in ("fp32", "emulated"):
                raise ValueError(f"precision must be 'fp32' or 'emulated', got {self.precision!r}")
            W = self[0].weight
            if self.precision == "emulated" and x.is_cuda and ops.encode_dense_emu_supported(W.shape[1]):
                if not hasattr(self, "_emu_cache"):
                    self._emu_cache = PackedCache()
                st = self._emu_cache.get((W,), lambda: dict(zip(("Wc", "meta2"), ops.emu_pack_w(W.detach()))))
                return ops.encode_dense_emu(x, st["Wc"], st["meta2"], self[0].bias, self._act)
            xp, Wp, kperm = self.operands(x)
            return ops.encode_dense(xp, Wp, self[0].bias, self._act, kperm=kperm)

    def operands(self, x: torch.Tensor):
        """(x', W', kperm): K-interleaved copies when the input width allows it (qsae_kperm_rows) -- W
        once per checkpoint (cached on the parameter's version), x per call -- else the originals."""
        W = self[0].weight
        if W.shape[1] % 32 != 0 or not x.is_cuda:
            return x, W, False
        if not hasattr(self, "_kperm_cache"):
            self._kperm_cache = PackedCache()
        Wp = self._kperm_cache.get((W,), lambda: {"Wp": ops.kperm_rows(W.detach())})["Wp"]
        return ops.kperm_rows(x), Wp, True


class PackedCache:
    """Derived (packed) weights keyed on the source parameters' identity and version counter, so
    that load_state_dict / .to(device) / in-place edits invalidate them automatically."""

    def __init__(self):
        self._key = None
        self._value = None

    @staticmethod
    def _sig(tensors) -> Tuple:
        return tuple((t.data_ptr(), t._version, str(t.device), tuple(t.shape)) for t in tensors)

    def get(self, tensors, build: Callable[[], Dict]):
        key = self._sig(tensors)
        if key != self._key:
            self._value = build()
            self._key = key
        return self._value

    def clear(self):
        self._key = None
        self._value = None


class SparseAutoencoder(nn.Module):
    """encode -> decode -> (latent, reconstruction); subclasses install encoder/decoder."""

    def __init__(self, input_dim: int, hidden_dim: int):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.encoder = None
        self.decoder = None

    def encode(self, x):
        if self.encoder is None:
            raise NotImplementedError("Encoder has not been implemented.")
        return self.encoder(x)

    def decode(self, h):
        if self.decoder is None:
            raise NotImplementedError("Decoder has not been implemented.")
        return self.decoder(h)

    def forward(self, x):
        latent = self.encode(x)
        return latent, self.decode(latent)


def require_device_input(x: torch.Tensor, what: str = "input") -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(x)}")
    if not x.is_cuda:
        raise RuntimeError(
            f"{what} is on {x.device}: quantizedsae_amd computes on MI355X only and has no CPU fallback; "
            "move the model and the batch to a ROCm device")
    if x.dim() != 2:
        raise ValueError(f"{what}: expected a [batch, features] tensor, got shape {tuple(x.shape)}")
    return x
