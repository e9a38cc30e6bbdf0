"""BinarySAE: fp32 encoder, top-k mask, n-bit two's-complement dictionary
(reference: sae/binary.py:10-103)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .base import HipEncoder, PackedCache, SparseAutoencoder, require_device_input


class binary_decoder(nn.Module):
    """Dictionary stored as per-bit logits ``weight[h, d*n_bits + b]`` (bit b of output d, LSB
    first, MSB weight negative), ``bias[d]`` (sae/binary.py:11-22).

    Inference uses the *hard* bits (``sigmoid(w) > 0.5``, i.e. ``quantized_int_weights()``,
    sae/binary.py:49-58) packed to n-bit fields.  The reference forward multiplies with the
    *soft* sigmoid bits; the two agree to ~1e-7 once the logits are polarised (|w| >= 20).
    ``decode_mode = "soft"`` reproduces the soft arithmetic with an fp32 table for
    unpolarised checkpoints.
    """

    def __init__(self, in_features, out_features, gamma=4.0, n_bits=8):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.n_bits = n_bits
        self.scale_factor = 2 ** n_bits
        self.gamma = gamma
        self.quantization_step = gamma / (2 ** (n_bits - 1))
        self.weight = nn.Parameter(torch.empty(in_features, out_features * n_bits))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.kaiming_normal_(self.weight)
        self.decode_mode = "hard"
        self._cache = PackedCache()

    # -- packed state -------------------------------------------------------------------------
    def packed(self) -> dict:
        """{'packed': uint8 [H,row_bytes], 'polarize': fp32 0-d, ['soft_table': fp32 [H,D]]}"""
        def build():
            w = require_device_input(self.weight.detach(), "decoder.weight")
            packed, pol = ops.pack_binary(w, self.out_features, self.n_bits)
            polarize = (pol / float(w.numel())).to(torch.float32)
            return {"packed": packed, "polarize": polarize}
        return self._cache.get((self.weight,), build)

    def soft_table(self) -> torch.Tensor:
        st = self.packed()
        if "soft_table" not in st:
            st["soft_table"] = ops.binary_soft_table(self.weight.detach(), self.out_features, self.n_bits)
        return st["soft_table"]

    # -- sparse decode (the hot path) -----------------------------------------------------------
    def decode_sparse(self, idx: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
        st = self.packed()
        if self.decode_mode == "soft":
            return ops.decode_table_sparse(idx, val, self.soft_table(), self.quantization_step, self.bias.detach())
        if self.decode_mode != "hard":
            raise ValueError(f"decode_mode must be 'hard' or 'soft', got {self.decode_mode!r}")
        return ops.decode_binary_sparse(idx, val, st["packed"], self.out_features, self.n_bits,
                                        self.quantization_step, self.bias.detach())

    # -- reference-compatible dense entry point ---------------------------------------------------
    def forward(self, latent, true_sum=None):
        """(reconstruction, polarize_loss) for an arbitrary dense latent [B, H]
        (sae/binary.py:24-47; ``true_sum`` is ignored there as well)."""
        with torch.no_grad():
            latent = require_device_input(latent, "latent")
            st = self.packed()
            table = self.soft_table() if self.decode_mode == "soft" else self._int_table()
            acc = ops.encode_dense(latent, table.t().contiguous(), None, ops.ACT_NONE)
            recon = self.quantization_step * acc + self.bias.detach()
            return recon, st["polarize"]

    def _int_table(self) -> torch.Tensor:
        return ops.unpack_binary(self.packed()["packed"], self.out_features, self.n_bits)

    def quantized_int_weights(self):
        """Two's-complement integer weights [H, D] as fp32 (sae/binary.py:49-58)."""
        with torch.no_grad():
            return self._int_table()

    def quantized_int_weights_continuous(self):
        """Soft (sigmoid-bit) integer weights [H, D] (sae/binary.py:60-69)."""
        with torch.no_grad():
            return ops.binary_soft_table(self.weight.detach(), self.out_features, self.n_bits)


class BinarySAE(SparseAutoencoder):
    """``forward(x) -> (sparse_latent [B,H], reconstruction [B,D], polarize_loss [])``
    (sae/binary.py:71-103).  k = int(hidden_dim * self.k) with self.k = 0.002."""

    def __init__(self, input_dim, hidden_dim, gamma=4.0, n_bits=8):
        super().__init__(input_dim, hidden_dim)
        self.n_bits = n_bits
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.k = 0.002
        lin = nn.Linear(input_dim, hidden_dim)
        nn.init.xavier_uniform_(lin.weight, gain=1)
        nn.init.zeros_(lin.bias)
        self.encoder = HipEncoder(lin)
        self.decoder = binary_decoder(hidden_dim, input_dim, gamma=gamma, n_bits=self.n_bits)

    @property
    def top_k(self) -> int:
        return int(self.hidden_dim * self.k)

    def forward_compact(self, x):
        """(idx int32 [B,k], val fp32 [B,k], reconstruction [B,D]) without the dense latent; same path
        selection (and the same bits) as forward()."""
        with torch.no_grad():
            x = require_device_input(x, "x")
            lin = self.encoder.linear
            if self.resolved_latent_path(x.shape[0]) == "prefilter":
                pw = self._prefilter_weights()
                xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
                if self.decoder.decode_mode == "hard" and self.fuse_decode:
                    dec = self.decoder
                    idx, val, _, recon = ops.binary_forward_prefilter(
                        xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, dec.packed()["packed"],
                        dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=False)
                    return idx, val, recon
                idx, val, _ = ops.encode_topk_prefilter(xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"],
                                                        self.top_k, want_dense=False)
            else:
                xp, Wp, kperm = self.encoder.operands(x)
                idx, val = ops.encode_topk(xp, Wp, lin.bias, self.top_k, kperm=kperm)
            return idx, val, self.decoder.decode_sparse(idx, val)

    #: "auto" | "fused" | "inplace" | "prefilter".  fused: exact-fp32 encoder+top-k without a dense
    #: latent in HBM (the dense [B,H] return value is zero-filled inside the sweep); inplace: dense
    #: contraction, top-k masks it in place; prefilter: an fp16 MFMA pass with a rigorous error bound
    #: picks ~90 candidates per row, which are re-evaluated and ranked in exact fp32.  All paths return
    #: bit-identical outputs; auto takes prefilter for large batches (fused where its shape limits do not
    #: hold) and inplace for small ones.
    latent_path = "auto"
    #: prefilter path: the refinement kernel also decodes each row it has ranked (one launch less, the decode's work in
    #: the refinement's idle issue slots); False = separate decode kernel.  Bit-identical either way.
    fuse_decode = True

    def resolved_latent_path(self, batch_rows: int) -> str:
        """Which path forward() takes for a batch of this many rows (after the auto / shape fallbacks)."""
        path = self.latent_path
        big = batch_rows >= 2048 and self.hidden_dim >= 8192
        if path == "auto":
            path = "prefilter" if big else "inplace"
        if path == "prefilter" and not ops.prefilter_supported(batch_rows, self.input_dim, self.hidden_dim, self.top_k):
            path = "fused" if big else "inplace"
        return path

    def _prefilter_weights(self):
        lin = self.encoder.linear
        if not hasattr(self, "_pref_cache"):
            self._pref_cache = PackedCache()
        def build():
            Wq, meta = ops.prefilter_pack_w(lin.weight.detach(), lin.bias.detach())
            return {"Wq": Wq, "meta": meta}
        return self._pref_cache.get((lin.weight, lin.bias), build)

    def forward(self, x):
        with torch.no_grad():
            x = require_device_input(x, "x")
            lin = self.encoder.linear
            path = self.latent_path
            if path == "auto":
                path = "prefilter" if (x.shape[0] >= 2048 and self.hidden_dim >= 8192) else "inplace"
            if path == "prefilter":
                if not ops.prefilter_supported(x.shape[0], self.input_dim, self.hidden_dim, self.top_k):
                    path = "fused" if (x.shape[0] >= 2048 and self.hidden_dim >= 8192) else "inplace"
                else:
                    pw = self._prefilter_weights()
                    xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
                    if self.decoder.decode_mode == "hard" and self.fuse_decode:
                        dec = self.decoder
                        st = dec.packed()
                        idx, val, latent, recon = ops.binary_forward_prefilter(
                            xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, st["packed"],
                            dec.n_bits, dec.quantization_step, dec.bias.detach())
                        return latent, recon, st["polarize"]
                    idx, val, latent = ops.encode_topk_prefilter(xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"],
                                                                 pw["meta"], self.top_k)
            if path == "fused":
                xp, Wp, kperm = self.encoder.operands(x)
                idx, val, latent = ops.encode_topk_latent(xp, Wp, lin.bias, self.top_k, kperm=kperm)
            elif path == "inplace":
                latent = self.encoder(x)
                idx, val = ops.topk_rows(latent, self.top_k, zero_rest=True)     # latent * mask, in place
            elif path != "prefilter":
                raise ValueError(f"latent_path must be 'auto', 'fused', 'inplace' or 'prefilter', got {path!r}")
            recon = self.decoder.decode_sparse(idx, val)
            return latent, recon, self.decoder.packed()["polarize"]
