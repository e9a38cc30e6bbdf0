"""ResidualQuantizedSAE: a chain of 1-bit matryoshka SAEs on doubling residuals
(reference: sae/residual_quantized.py:11-74)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import SparseAutoencoder, require_device_input
from .quantized_matryoshka import QuantizedMatryoshkaSAE, nested_sizes


class ResidualQuantizedSAE(ops.GraphForwardMixin, SparseAutoencoder):
    """``forward(x) -> (latent_groups, reconstruction_levels)``: stage i encodes the residual left
    by stage i-1, ``residual = (residual - recon) * 2``; only stage 0 has a decoder bias."""

    def __init__(self, input_dim, hidden_dim, top_k, abs_range=4, n_bits=8):
        super().__init__(input_dim, hidden_dim)
        self.n_bits = n_bits
        self.abs_range = abs_range
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.top_k = top_k
        self.sae_hidden_dims = nested_sizes(hidden_dim, n_bits)
        self.saes = nn.ModuleList([
            QuantizedMatryoshkaSAE(input_dim=input_dim, hidden_dim=h, top_k=top_k, abs_range=abs_range, n_bits=1,
                                   allow_bias=(i == 0))
            for i, h in enumerate(self.sae_hidden_dims)
        ])
        ops.module_handle(self)

    def forward(self, x):
        if torch.compiler.is_compiling():                  # one graph node: torch.ops.qsae.levels_sae_forward
            with torch.no_grad():
                groups, levels = torch.ops.qsae.levels_sae_forward(x, [p for p in self.parameters()], self._qsae_handle)
            return [groups[i] for i in range(self.n_bits)], [levels[i] for i in range(self.n_bits)]
        return self._forward_eager(x)

    def _forward_eager(self, x):
        with torch.no_grad():
            residual = require_device_input(x, "x")
            if residual.dtype != torch.float32:
                residual = residual.float()
            groups, levels = [], []
            for sae in self.saes:
                g, recs = sae(residual)
                groups.append(g[-1])
                levels.append(recs[-1])
                residual = ops.residual_update(residual, recs[-1], 2.0)          # (residual - reconstruction) * 2
            return groups, levels

    def apply_secant_grad(self):
        for sae in self.saes:
            sae.decoder.apply_secant_grad()
