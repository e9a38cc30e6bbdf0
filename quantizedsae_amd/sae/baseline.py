"""BaselineSparseAutoencoder: Linear -> top-32 -> Linear (reference: sae/baseline.py:4-51)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import HipEncoder, PackedCache, require_device_input


class BaselineSparseAutoencoder(ops.GraphForwardMixin, nn.Module):
    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.encoder = HipEncoder(nn.Linear(input_dim, hidden_dim))   # no ReLU in the reference either
        self.decoder = nn.Linear(hidden_dim, input_dim)
        self.topk = 32
        self.latent_path = "auto"      # "auto" | "prefilter" | "fused" | "inplace" (see BinarySAE.latent_path)
        self._cache = PackedCache()
        self._pref_cache = PackedCache()
        self.last_flagged_rows = 0     # rows of the previous prefilter batch that took the exact fallback (per model)
        ops.module_handle(self)

    def _table(self) -> torch.Tensor:
        # decoder.weight is [D, H]; the sparse decode gathers rows of its transpose [H, D]
        return self._cache.get((self.decoder.weight,),
                               lambda: {"t": self.decoder.weight.detach().t().contiguous()})["t"]

    def _prefilter_ok(self, rows: int) -> bool:
        lin = self.encoder.linear
        H, D = lin.weight.shape
        big = rows >= 2048 and H >= 8192
        return big and self.latent_path in ("auto", "prefilter") and ops.prefilter_supported(rows, D, H, self.topk)

    def _prefilter_weights(self):
        lin = self.encoder.linear
        return self._pref_cache.get((lin.weight, lin.bias), lambda: dict(zip(
            ("Wq", "meta"), ops.prefilter_pack_w(lin.weight.detach(), lin.bias.detach()))))

    def _run(self, x, want_dense: bool):
        """-> (idx, val, dense latent or None, reconstruction): the one implementation behind forward() and
        forward_compact()."""
        x = require_device_input(x, "x")
        lin = self.encoder.linear
        H = lin.weight.shape[0]
        big = x.shape[0] >= 2048 and H >= 8192
        if self._prefilter_ok(x.shape[0]):
            # one call: candidate sweep, exact refinement, and the row's reconstruction from the fp32 decoder rows as soon
            # as the row is ranked (qsae_table_forward_prefilter)
            pw = self._prefilter_weights()
            xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
            info = {}
            idx, val, h, recon = ops.table_forward_prefilter(
                xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.topk, self._table(), 1.0,
                self.decoder.bias.detach(), want_dense=want_dense, spec_rows=32 if self.last_flagged_rows > 0 else 0,
                info=info)
            self.last_flagged_rows = info["flagged_rows"]
            return idx, val, h, recon
        if big and self.latent_path != "inplace" and want_dense:
            xp, Wp, kperm = self.encoder.operands(x)
            idx, val, h = ops.encode_topk_latent(xp, Wp, lin.bias, self.topk, kperm=kperm)
        elif not want_dense:
            xp, Wp, kperm = self.encoder.operands(x)
            idx, val = ops.encode_topk(xp, Wp, lin.bias, self.topk, kperm=kperm)
            h = None
        else:
            h = self.encoder(x)
            idx, val = ops.topk_rows(h, self.topk, zero_rest=True)
        return idx, val, h, ops.decode_table_sparse(idx, val, self._table(), 1.0, self.decoder.bias.detach())

    def forward(self, x):
        """-> (h_sparse [B,H], recon [B,D])  (sae/baseline.py:17-31)."""
        with torch.no_grad():
            if torch.compiler.is_compiling():          # one graph node: torch.ops.qsae.baseline_sae_forward
                lin = self.encoder.linear
                _, _, h, recon = torch.ops.qsae.baseline_sae_forward(
                    x, [lin.weight, lin.bias, self.decoder.weight, self.decoder.bias], self._qsae_handle, True)
                return h, recon
            _, _, h, recon = self._run(x, want_dense=True)
            return h, recon

    def forward_compact(self, x):
        """(idx, val, reconstruction) without the dense latent; same path selection as forward()."""
        with torch.no_grad():
            idx, val, _, recon = self._run(x, want_dense=False)
            return idx, val, recon

    def forward_submit(self, x, slot: int = 0, want_dense: bool = True):
        """Queue one forward without waiting for the GPU anywhere (see BinarySAE.forward_submit): ``result()`` of the
        returned handle gives ``(h_sparse, reconstruction)`` (``(idx, val, reconstruction)`` with want_dense=False)."""
        with torch.no_grad():
            xd = require_device_input(x, "x")
            if self._prefilter_ok(xd.shape[0]):
                lin = self.encoder.linear
                pw = self._prefilter_weights()
                xf = xd if (xd.dtype == torch.float32 and xd.is_contiguous()) else xd.float().contiguous()
                pending = ops.table_forward_prefilter_submit(
                    xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.topk, self._table(), 1.0,
                    self.decoder.bias.detach(), want_dense=want_dense, slot=slot, owner=self._qsae_handle)
                return _SubmittedBaseline(self, pending, None, want_dense)
            return _SubmittedBaseline(self, None, self._run(xd, want_dense), want_dense)

    def invalidate_packed(self) -> None:
        """Forget the derived weight copies (transposed decoder table, fp16 / K-interleaved encoder copies): needed only
        after an in-place edit through ``.data`` -- normalize_decoder_weights() below does it itself."""
        self._cache.clear()
        self._pref_cache.clear()
        if hasattr(self.encoder, "_kperm_cache"):
            self.encoder._kperm_cache.clear()

    def apply_topk_activation(self, h):
        """Dense in, dense out: keep the top-k entries of every row (sae/baseline.py:33-40)."""
        with torch.no_grad():
            out = require_device_input(h, "h").float().clone()
            ops.topk_rows(out, self.topk, zero_rest=True)
            return out

    def normalize_decoder_weights(self):
        """Unit-norm decoder columns (sae/baseline.py:42-51; training utility, plain torch ops)."""
        with torch.no_grad():
            w = self.decoder.weight.data
            self.decoder.weight.data = w / torch.clamp(torch.norm(w, dim=0, keepdim=True), min=1e-8)
            self.invalidate_packed()


class _SubmittedBaseline:
    def __init__(self, model, pending, outs, want_dense):
        self._model, self._pending, self._outs, self._want_dense = model, pending, outs, want_dense

    def result(self):
        with torch.no_grad():
            if self._pending is not None:
                self._outs = self._pending.finish()
                self._model.last_flagged_rows = self._pending.flagged_rows
                self._pending = None
            idx, val, h, recon = self._outs
            return (h, recon) if self._want_dense else (idx, val, recon)
