"""BinaryLatentSAE: sigmoid encoder, latent binarised at 0.5, dense fp32 decoder
(reference: sae/binary_latent.py:6-28)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import HipEncoder, SparseAutoencoder, require_device_input

# sigmoid(w) >= 0.5 in the reference's fp32 op sequence  <=>  w >= this pre-activation (bit pattern 0xB43FFFFE,
# -1.79e-7; measured over all floats, tests/golden/sigmoid_cutoffs.npz)
_GE_HALF_CUTOFF = float(torch.tensor([0xB43FFFFE - (1 << 32)], dtype=torch.int32).view(torch.float32)[0])


class BinaryLatentSAE(SparseAutoencoder):
    """``forward(x) -> (binary_latent [B,H] in {0,1}, reconstruction [B,D])``.  The reference decodes
    ``latent + (binary - latent).detach()``, which is the binary latent up to one rounding (<= 6e-8); here the
    decoder contracts the binary latent itself."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__(input_dim, hidden_dim)
        self.encoder = HipEncoder(nn.Linear(input_dim, hidden_dim), nn.Sigmoid())
        self.decoder = nn.Linear(hidden_dim, input_dim)

    def forward(self, x):
        with torch.no_grad():
            x = require_device_input(x, "x")
            lin = self.encoder.linear
            pre = ops.encode_dense(x if x.dtype == torch.float32 else x.float(), lin.weight.detach(), lin.bias.detach(),
                                   ops.ACT_NONE)
            binary_latent = ops.threshold_ge(pre, _GE_HALF_CUTOFF)
            recon = ops.encode_dense(binary_latent, self.decoder.weight.detach(), self.decoder.bias.detach(),
                                     ops.ACT_NONE)
            return binary_latent, recon
