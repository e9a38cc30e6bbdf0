"""BinarySAE: fp32 encoder, top-k mask, n-bit two's-complement dictionary
(reference: sae/binary.py:10-103)."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import HipEncoder, PackedCache, SparseAutoencoder, require_device_input


class binary_decoder(nn.Module):
    """Dictionary stored as per-bit logits ``weight[h, d*n_bits + b]`` (bit b of output d, LSB
    first, MSB weight negative), ``bias[d]`` (sae/binary.py:11-22).

    The reference forward multiplies the latent with the *soft* integers ``sum_b sigmoid(w_b) bw_b``
    (sae/binary.py:26-38); the *hard* two's-complement integers (``sigmoid(w) > 0.5``,
    ``quantized_int_weights()``, sae/binary.py:49-58) are what a trained, polarised checkpoint converges to
    and what the packed n-bit dictionary holds.  Which of the two a forward uses is decided when the
    dictionary is packed, from a number the packer measures on the way:

        soft_gap = max over (h, d) of |soft integer - hard integer|        (qsae_pack_binary)

    ``decode_mode``:
      * ``"auto"`` (default) -- hard packed decode when ``soft_gap <= hard_max_gap * (2^n_bits - 1)`` (1e-7 of the
        integer range: every logit beyond about +-16.2, whatever n_bits; reconstructions then equal the reference's to
        ~1e-6 relative), otherwise the reference's soft
        arithmetic over an fp32 ``[H, D]`` table (a warning says so once per checkpoint);
      * ``"hard"`` -- always the packed integers (the deployment form; differs from the reference forward
        on an unpolarised checkpoint by up to tens of percent, see scripts/evaluation/estimate_quantization_error.py);
      * ``"soft"`` -- always the soft table.
    """

    #: auto mode: largest |soft - hard|, as a fraction of the integer range 2^n_bits - 1 (a per-bit saturation measure:
    #: the gap of a dictionary whose logits all sit at +-L is (2^n_bits - 1) sigmoid(-L)), for which the hard decode
    #: stands in for the soft one
    hard_max_gap = 1e-7

    def hard_gap_limit(self) -> float:
        return self.hard_max_gap * float(2 ** self.n_bits - 1)

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
        self.decode_mode = "auto"
        self._cache = PackedCache()

    # -- packed state -------------------------------------------------------------------------
    def packed(self) -> dict:
        """{'packed': uint8 [H,row_bytes], 'polarize': fp32 0-d, 'soft_gap': float, ['soft_table': fp32 [H,D]]}"""
        def build():
            w = require_device_input(self.weight.detach(), "decoder.weight")
            packed, pol, gap = ops.pack_binary(w, self.out_features, self.n_bits, want_soft_gap=True)
            polarize = (pol / float(w.numel())).to(torch.float32)
            return {"packed": packed, "polarize": polarize, "soft_gap": float(gap)}   # one host read per checkpoint
        return self._cache.get((self.weight,), build)

    def invalidate_packed(self) -> None:
        """Forget the packed dictionary.  Needed only after an in-place edit THROUGH ``.data`` (``w.data.mul_()``),
        which bumps no version counter; ``load_state_dict``, ``.to()``, optimizer steps and ``no_grad`` in-place ops
        on the parameter itself are noticed automatically."""
        self._cache.clear()

    def soft_table(self) -> torch.Tensor:
        st = self.packed()
        if "soft_table" not in st:
            st["soft_table"] = ops.binary_soft_table(self.weight.detach(), self.out_features, self.n_bits)
        return st["soft_table"]

    def resolved_decode_mode(self) -> str:
        """"hard" or "soft": what decode_mode means for the current weights."""
        mode = self.decode_mode
        if mode in ("hard", "soft"):
            return mode
        if mode != "auto":
            raise ValueError(f"decode_mode must be 'auto', 'hard' or 'soft', got {mode!r}")
        st = self.packed()
        if st["soft_gap"] <= self.hard_gap_limit():
            return "hard"
        if not st.get("warned"):
            st["warned"] = True
            warnings.warn(
                f"binary_decoder: checkpoint is not polarised (max |soft - hard| integer gap {st['soft_gap']:.3g} > "
                f"{self.hard_gap_limit():g}); decoding with the reference's soft sigmoid-bit table (sae/binary.py:26-38). "
                "Set decode_mode='hard' for the packed two's-complement dictionary.", stacklevel=3)
        return "soft"

    # -- sparse decode (the hot path) -----------------------------------------------------------
    def decode_sparse(self, idx: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
        if self.resolved_decode_mode() == "soft":
            return ops.decode_table_sparse(idx, val, self.soft_table(), self.quantization_step, self.bias.detach())
        return ops.decode_binary_sparse(idx, val, self.packed()["packed"], self.out_features, self.n_bits,
                                        self.quantization_step, self.bias.detach())

    # -- reference-compatible dense entry point ---------------------------------------------------
    def forward(self, latent, true_sum=None):
        """(reconstruction, polarize_loss) for an arbitrary dense latent [B, H]
        (sae/binary.py:24-47; ``true_sum`` is ignored there as well)."""
        with torch.no_grad():
            latent = require_device_input(latent, "latent")
            st = self.packed()
            table = self.soft_table() if self.resolved_decode_mode() == "soft" else self._int_table()
            acc = ops.encode_dense(latent, table.t().contiguous(), None, ops.ACT_NONE)
            recon = ops.scale_bias_rows(acc, self.quantization_step, self.bias.detach())
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


class BinarySAE(ops.GraphForwardMixin, SparseAutoencoder):
    """``forward(x) -> (sparse_latent [B,H], reconstruction [B,D], polarize_loss [])``
    (sae/binary.py:71-103).  k = int(hidden_dim * self.k) with self.k = 0.002.

    Shape limits of the kernels (checked at the first forward with a clear message): hidden_dim <= 32768 for the
    in-place path and the exact fallback, k <= 256.  k == 0 (hidden_dim < 500) is served like the reference: an
    all-zero latent and a bias-only reconstruction."""

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
        self._pref_cache = PackedCache()
        #: rows of the previous prefilter batch that went through the exact fallback kernels (sizes the next call's
        #: speculative fallback; per model, not per process)
        self.last_flagged_rows = 0
        ops.module_handle(self)

    @property
    def top_k(self) -> int:
        return int(self.hidden_dim * self.k)

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
        if path not in ("auto", "fused", "inplace", "prefilter"):
            raise ValueError(f"latent_path must be 'auto', 'fused', 'inplace' or 'prefilter', got {path!r}")
        big = batch_rows >= 2048 and self.hidden_dim >= 8192
        if path == "auto":
            path = "prefilter" if big else "inplace"
        if path == "prefilter" and not ops.prefilter_supported(batch_rows, self.input_dim, self.hidden_dim, self.top_k):
            path = "fused" if big else "inplace"
        return path

    def _prefilter_weights(self):
        lin = self.encoder.linear
        def build():
            Wq, meta = ops.prefilter_pack_w(lin.weight.detach(), lin.bias.detach())
            return {"Wq": Wq, "meta": meta}
        return self._pref_cache.get((lin.weight, lin.bias), build)

    def invalidate_packed(self) -> None:
        """Forget every derived copy of the weights (packed dictionary, fp16 / K-interleaved encoder copies); see
        binary_decoder.invalidate_packed."""
        self.decoder.invalidate_packed()
        self._pref_cache.clear()
        if hasattr(self.encoder, "_kperm_cache"):
            self.encoder._kperm_cache.clear()

    def _check_limits(self, path: str) -> None:
        k = self.top_k
        if k > 256:
            raise ValueError(f"BinarySAE: top-k = int({self.hidden_dim} * {self.k}) = {k} exceeds the kernels' limit of 256")
        if self.hidden_dim > 32768 and path == "inplace":
            raise ValueError(f"BinarySAE: hidden_dim = {self.hidden_dim} exceeds the in-place top-k kernel's limit of 32768 "
                             "(batches of >= 2048 rows take the fused path up to 65536)")

    def _zero_k(self, x, want_dense: bool):
        """k == 0: the reference's topk(0) keeps nothing -- zero latent, reconstruction = decoder bias (sae/binary.py:94-99)."""
        B = x.shape[0]
        latent = torch.zeros((B, self.hidden_dim), dtype=torch.float32, device=x.device) if want_dense else None
        recon = self.decoder.bias.detach().to(torch.float32).expand(B, -1).contiguous()
        idx = torch.empty((B, 0), dtype=torch.int32, device=x.device)
        val = torch.empty((B, 0), dtype=torch.float32, device=x.device)
        return idx, val, latent, recon

    def _run(self, x, want_dense: bool):
        """-> (idx, val, dense latent or None, reconstruction): the one implementation behind forward() and
        forward_compact()."""
        x = require_device_input(x, "x")
        if self.top_k == 0:
            return self._zero_k(x, want_dense)
        lin = self.encoder.linear
        path = self.resolved_latent_path(x.shape[0])
        self._check_limits(path)
        hard = self.decoder.resolved_decode_mode() == "hard"
        latent = None
        if path == "prefilter":
            pw = self._prefilter_weights()
            xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
            spec = 32 if self.last_flagged_rows > 0 else 0
            info = {}
            if self.fuse_decode:
                # one call: the refinement kernel decodes every row it ranks -- from the packed n-bit dictionary, or from
                # the fp32 soft-integer table when the checkpoint is not polarised (the reference's own arithmetic)
                dec = self.decoder
                if hard:
                    idx, val, latent, recon = ops.binary_forward_prefilter(
                        xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, dec.packed()["packed"],
                        dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=want_dense, spec_rows=spec, info=info)
                else:
                    idx, val, latent, recon = ops.table_forward_prefilter(
                        xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, dec.soft_table(),
                        dec.quantization_step, dec.bias.detach(), want_dense=want_dense, spec_rows=spec, info=info)
                self.last_flagged_rows = info["flagged_rows"]
                return idx, val, latent, recon
            idx, val, latent = ops.encode_topk_prefilter(xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"],
                                                         self.top_k, want_dense=want_dense, spec_rows=spec, info=info)
            self.last_flagged_rows = info["flagged_rows"]
        elif path == "fused" or not want_dense:
            xp, Wp, kperm = self.encoder.operands(x)
            if want_dense:
                idx, val, latent = ops.encode_topk_latent(xp, Wp, lin.bias, self.top_k, kperm=kperm)
            else:
                idx, val = ops.encode_topk(xp, Wp, lin.bias, self.top_k, kperm=kperm)
        else:   # inplace
            latent = self.encoder(x)
            idx, val = ops.topk_rows(latent, self.top_k, zero_rest=True)     # latent * mask, in place
        return idx, val, latent, self.decoder.decode_sparse(idx, val)

    def _graph_params(self):
        lin, dec = self.encoder.linear, self.decoder
        return [lin.weight, lin.bias, dec.weight, dec.bias]

    def forward_compact(self, x):
        """(idx int32 [B,k], val fp32 [B,k], reconstruction [B,D]) without the dense latent; same path
        selection (and the same bits) as forward()."""
        with torch.no_grad():
            if torch.compiler.is_compiling():          # one graph node: torch.ops.qsae.binary_sae_forward
                idx, val, _, recon, _ = torch.ops.qsae.binary_sae_forward(x, self._graph_params(), self._qsae_handle, False)
                return idx, val, recon
            idx, val, _, recon = self._run(x, want_dense=False)
            return idx, val, recon

    def forward(self, x):
        with torch.no_grad():
            if torch.compiler.is_compiling():          # one graph node: torch.ops.qsae.binary_sae_forward
                _, _, latent, recon, pol = torch.ops.qsae.binary_sae_forward(x, self._graph_params(), self._qsae_handle, True)
                return latent, recon, pol
            _, _, latent, recon = self._run(x, want_dense=True)
            return latent, recon, self.decoder.packed()["polarize"]

    # -- two batches in flight --------------------------------------------------------------------------------
    def forward_submit(self, x, slot: int = 0, want_dense: bool = True):
        """Queue one forward without waiting for the GPU anywhere: returns a handle whose ``result()`` gives
        ``(latent, reconstruction, polarize_loss)`` (``(idx, val, reconstruction)`` with want_dense=False).  With the
        default path this is the two-call form of the C ABI (qsae_prefilter_submit / _finish, or their _table forms for
        an unpolarised checkpoint): submit batch i+1, then
        call ``result()`` of batch i -- the 4-byte read-back of batch i no longer idles the GPU.  Batches in flight
        together need different ``slot`` numbers; other paths compute eagerly and return a finished handle."""
        with torch.no_grad():
            xd = require_device_input(x, "x")
            if self.top_k > 0 and self.resolved_latent_path(xd.shape[0]) == "prefilter" and self.fuse_decode:
                self._check_limits("prefilter")
                lin, dec = self.encoder.linear, self.decoder
                pw = self._prefilter_weights()
                xf = xd if (xd.dtype == torch.float32 and xd.is_contiguous()) else xd.float().contiguous()
                if dec.resolved_decode_mode() == "hard":
                    pending = ops.binary_forward_prefilter_submit(
                        xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, dec.packed()["packed"],
                        dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=want_dense, slot=slot, owner=self._qsae_handle)
                else:
                    pending = ops.table_forward_prefilter_submit(
                        xf, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], self.top_k, dec.soft_table(),
                        dec.quantization_step, dec.bias.detach(), want_dense=want_dense, slot=slot, owner=self._qsae_handle)
                return _SubmittedForward(self, pending, None, want_dense)
            return _SubmittedForward(self, None, self._run(xd, want_dense), want_dense)


class _SubmittedForward:
    def __init__(self, model, pending, outs, want_dense):
        self._model, self._pending, self._outs, self._want_dense = model, pending, outs, want_dense

    def result(self):
        with torch.no_grad():
            if self._pending is not None:
                self._outs = self._pending.finish()
                self._model.last_flagged_rows = self._pending.flagged_rows
                self._pending = None
            idx, val, latent, recon = self._outs
            if self._want_dense:
                return latent, recon, self._model.decoder.packed()["polarize"]
            return idx, val, recon
