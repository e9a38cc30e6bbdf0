"""TernarySparseAutoencoder: ReLU encoder, {-1,0,+1} dictionary (reference: sae/ternary.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import HipEncoder, PackedCache, require_device_input

_TRAINING_ONLY = ("RigL mask maintenance is part of the reference's training loop "
                  "(sae/ternary.py:27-39,54-90) and is outside this inference backend")


class STEWeights(nn.Module):
    """Ternary dictionary ``hard = sign(w) * (|w| >= threshold)``, no bias (sae/ternary.py:41-52).
    The straight-through expression of the reference evaluates to exactly ``hard`` and does not
    depend on ``mask``; the 2-bit packed codes are derived from ``weight`` alone."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.threshold = 0.5
        self.register_buffer("mask", torch.ones(out_features, in_features))
        self.input_activations = None   # the reference pins the last [B,H] input here; not kept
        self.output_grad = None
        nn.init.kaiming_normal_(self.weight)
        self._cache = PackedCache()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # checkpoints saved by the reference after a forward carry these hook buffers
        for name in ("input_activations", "output_grad"):
            state_dict.pop(prefix + name, None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def codes(self) -> torch.Tensor:
        return self._packed()["c"]

    #: "auto" | "split" | "fp32".  split: the latent as three exact bf16 terms against the {-1, 0, +1} dictionary on the
    #: bf16 matrix pipe, fp32 accumulation (qsae_decode_ternary_dense_split) -- every product exact, the sum rounded in
    #: another order than the fp32 kernel's (both within 1e-5 of the fp64-accumulating oracle); fp32: the exact-fp32 MFMA
    #: chain.  auto = split where the kernel covers the shape (input_dim 512, hidden_dim % 64 == 0), else fp32.
    precision = "auto"

    def _packed(self) -> dict:
        if self.threshold != 0.5:
            raise NotImplementedError("only the reference threshold 0.5 is packed")
        return self._cache.get((self.weight,), lambda: {"c": ops.pack_ternary(self.weight.detach())})

    def dictionary_bf16(self) -> torch.Tensor:
        """The bf16 image of the dictionary the split kernel streams (once per checkpoint, 2 H D bytes)."""
        st = self._packed()
        if "tq" not in st:
            D, H = self.weight.shape
            st["tq"] = ops.expand_codes_bf16(st["c"], D, H)
        return st["tq"]

    def resolved_precision(self, rows: int) -> str:
        if self.precision not in ("auto", "split", "fp32"):
            raise ValueError(f"precision must be 'auto', 'split' or 'fp32', got {self.precision!r}")
        D, H = self.weight.shape
        ok = rows > 0 and ops.split_dec_supported(rows, H, D)
        if self.precision == "split" and not ok:
            raise ValueError(f"STEWeights: the bf16 split decoder needs input_dim 512 and hidden_dim % 64 == 0 (got {D}, {H})")
        return "split" if (ok and self.precision != "fp32") else "fp32"

    def forward(self, x):
        with torch.no_grad():
            x = require_device_input(x, "x")
            if x.dtype != torch.float32 or x.stride(1) != 1:
                x = x.float().contiguous()
            if self.resolved_precision(x.shape[0]) == "split" and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
                return ops.decode_ternary_dense_split(x, self.dictionary_bf16(), self.weight.shape[0])
            return ops.decode_ternary_dense(x, self.codes(), self.weight.shape[0])

    def init_mask(self, sparsity):
        raise NotImplementedError(_TRAINING_ONLY)

    def update_mask(self, f_decay, sparsity_rate=0.7):
        raise NotImplementedError(_TRAINING_ONLY)

    def mask_grad(self):
        raise NotImplementedError(_TRAINING_ONLY)


class TernarySparseAutoencoder(ops.GraphForwardMixin, nn.Module):
    """``forward(x) -> (h [B,H], recon [B,D])``; no top-k in forward (sae/ternary.py:116-122)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.encoder = HipEncoder(nn.Linear(input_dim, hidden_dim), nn.ReLU())
        self.decoder = STEWeights(hidden_dim, input_dim)
        self.topk = int(hidden_dim * 0.002)
        ops.module_handle(self)

    def apply_topk_activation(self, h):
        """Top-k of each row with non-positive survivors zeroed (sae/ternary.py:102-114)."""
        with torch.no_grad():
            out = require_device_input(h, "h").float().clone()
            ops.topk_rows(out, self.topk, zero_rest=True)
            return torch.clamp_(out, min=0)

    def _forward_eager(self, x):
        h = self.encoder(require_device_input(x, "x"))
        return h, self.decoder(h)

    def forward(self, x):
        if torch.compiler.is_compiling():              # one graph node: torch.ops.qsae.ternary_sae_forward
            with torch.no_grad():
                lin = self.encoder.linear
                return torch.ops.qsae.ternary_sae_forward(x, [lin.weight, lin.bias, self.decoder.weight], self._qsae_handle)
        return self._forward_eager(x)
