"""QuantizedMatryoshkaSAE: sigmoid encoder binarised at 0.5, nested {-2,0,+2} dictionaries with
per-row scales (reference: sae/quantized_matryoshka.py:10-220)."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from .base import HipEncoder, PackedCache, SparseAutoencoder, require_device_input

_TRAINING_ONLY = ("the secant-gradient correction belongs to the reference's training loop "
                  "(sae/quantized_matryoshka.py:145-190) and is outside this inference backend")


def nested_sizes(in_features: int, n_bits: int) -> List[int]:
    """Level sizes [1,1,2,4,...] scaled to in_features, remainder in the last level
    (sae/quantized_matryoshka.py:25-38)."""
    sizes = [1 if i < 2 else 2 ** (i - 1) for i in range(n_bits)]
    total = sum(sizes)
    if total != in_features:
        f = in_features / total
        sizes = [max(1, int(s * f)) for s in sizes]
        sizes[-1] = in_features - sum(sizes[:-1])
    return sizes


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


class QuantizedMatryoshkaDecoder(nn.Module):
    """``forward(latent [B,H]) -> (latent_group: list of n 0-d tensors, result: list of n [B,D])``.

    Level i covers a slice of the hidden units; S = sgn(sigmoid(w) >= .5) + sgn(sigmoid(wm) >= .5),
    scale_j = 2^(n-i-2) * quant_step / (||S_j|| + 1e-8), z = latent > 0.5,
    recon_i = recon_{i-1} + (scale * z) @ S (+ bias once, after level 0).

    The packed form keeps S/2 as 2-bit fields (hidden index contiguous) and fp32 scales; the kernels
    need every level boundary on a multiple of 32, so odd level sizes are padded with inert units
    (S = 0, z = 0) at pack time.
    """

    def __init__(self, in_features, out_features, abs_range=4, n_bits=8, top_k=None, joint_gradient=False,
                 allow_bias=True):
        super().__init__()
        self._ctx = [None] * n_bits
        self.joint_gradient = joint_gradient
        self.in_features = in_features
        self.out_features = out_features
        self.n_bits = n_bits
        self.abs_range = abs_range
        self.quant_step = abs_range / (2 ** (n_bits - 1))
        self.top_k = top_k
        self.allow_bias = allow_bias
        self.nested_dictionary_size = nested_sizes(in_features, n_bits)
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.weight_mirror = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.xavier_uniform_(self.weight)
        nn.init.xavier_uniform_(self.weight_mirror)
        self._cache = PackedCache()

    # -- layout ---------------------------------------------------------------------------------
    @property
    def padded_sizes(self) -> List[int]:
        return [_pad32(s) for s in self.nested_dictionary_size]

    @property
    def needs_padding(self) -> bool:
        return self.padded_sizes != list(self.nested_dictionary_size)

    def padded_index(self, device) -> torch.Tensor:
        """For every padded hidden slot the source hidden unit, or -1 for an inert pad slot."""
        out = []
        start = 0
        for s, p in zip(self.nested_dictionary_size, self.padded_sizes):
            out.append(torch.arange(start, start + s, device=device))
            if p > s:
                out.append(torch.full((p - s,), -1, device=device, dtype=torch.long))
            start += s
        return torch.cat(out)

    def packed(self) -> dict:
        def build():
            w = require_device_input(self.weight.detach(), "decoder.weight")
            wm = self.weight_mirror.detach()
            sizes = list(self.nested_dictionary_size)
            st = {"sizes": sizes, "H": self.in_features, "index": None}
            if self.needs_padding:
                index = self.padded_index(w.device)
                valid = index >= 0
                Hp = int(index.numel())
                wp = torch.ones((Hp, self.out_features), device=w.device)
                wmp = -torch.ones((Hp, self.out_features), device=w.device)   # S = 0 on pad rows
                wp[valid] = w[index[valid]]
                wmp[valid] = wm[index[valid]]
                w, wm = wp, wmp
                st.update(sizes=self.padded_sizes, H=Hp, index=index)
            codes, scale = ops.pack_matryoshka(w, wm, self.n_bits, self.abs_range, st["sizes"])
            st.update(codes=codes, scale=scale)
            if ops.decode_matryoshka_sparse_supported(self.out_features):
                st["codes_rows"] = ops.pack_matryoshka_rows(w, wm)      # hidden-major copy for the sparse walk
            ends = [sum(st["sizes"][:i + 1]) for i in range(self.n_bits)]
            if ops.split_dec_supported(1, st["H"], self.out_features) and all(e % 64 == 0 for e in ends):
                st["tq"] = ops.expand_codes_bf16(codes, self.out_features, st["H"])      # bf16 image for the split kernel
                st["s3"] = ops.split_scale_bf16(scale)                                    # three bf16 terms of 2 scale
            return st
        return self._cache.get((self.weight, self.weight_mirror), build)

    # -- decode ---------------------------------------------------------------------------------
    #: "auto" | "fp32": the dense decoder on the bf16 matrix pipe where the kernel covers the shape (out_features 512,
    #: padded hidden size and level boundaries % 64 == 0), or always the exact-fp32 MFMA chain.  See STEWeights.precision.
    precision = "auto"
    #: the sparse walk beats the dense contraction below ~12 % active units (measured: 15 ms at 16 %, 23 ms dense)
    SPARSE_MAX_ACTIVE_FRACTION = 0.12

    def active_fraction_hint(self):
        """Fraction of active units in the most recent decode call whose counts have reached the host (None before the
        first one).  The counts travel to a page-locked buffer behind the decode (asynchronous copy + event) and are
        taken from there once the event has completed -- nothing here waits for the GPU, so a forward queued behind
        another one is not held up by the hint; at worst the hint is one batch older."""
        pending = getattr(self, "_pending_counts", None)
        if pending is not None:
            host, event, rows, units = pending
            if event.query():
                self._active_fraction = float(host.sum()) / max(rows * units, 1)
                self._pending_counts = None
        return getattr(self, "_active_fraction", None)

    def decode_bits(self, zbits: torch.Tensor, sparse=None) -> Tuple[list, list]:
        """zbits: int32-packed [B, H_padded/32] in the packed (padded) hidden order.  ``sparse``: walk the active
        units only (same outputs); None = decide from the previous call's activation density."""
        st = self.packed()
        B = zbits.shape[0]
        hint = self.active_fraction_hint()
        if hint is None and sparse is None and B > 0:
            # first decode of this model: measure this batch's density now (one host read, once per model) instead of
            # guessing -- the two decoders round their sums in different orders, and a model that switched after its
            # first call would return different low-order bits for the same batch
            hint = self._active_fraction = float(ops.activation_counts_bits(zbits).sum().item()) / max(B * st["H"], 1)
        if sparse is None:
            sparse = hint is not None and hint < self.SPARSE_MAX_ACTIVE_FRACTION
        elif sparse and hint is not None and hint >= self.SPARSE_MAX_ACTIVE_FRACTION:
            sparse = False
        if sparse and "codes_rows" in st:
            levels, counts = ops.decode_matryoshka_sparse(zbits, st["H"], self.out_features, self.n_bits,
                                                          st["codes_rows"], st["scale"], self.bias.detach(),
                                                          self.allow_bias, st["sizes"])
        elif self.precision != "fp32" and "tq" in st and B * zbits.stride(0) * 4 < (1 << 32):
            # dense activations: z_j * 2 scale_j as three exact bf16 terms against the {-1, 0, +1} dictionary on the bf16
            # matrix pipe (fp32 accumulation; qsae_decode_matryoshka_split)
            levels, counts = ops.decode_matryoshka_split(zbits, st["H"], self.out_features, self.n_bits, st["tq"], st["s3"],
                                                         self.bias.detach(), self.allow_bias, st["sizes"])
        else:
            levels, counts = ops.decode_matryoshka(zbits, st["H"], self.out_features, self.n_bits, st["codes"],
                                                   st["scale"], self.bias.detach(), self.allow_bias, st["sizes"])
        host = torch.empty((self.n_bits,), dtype=torch.int64).pin_memory()
        host.copy_(counts, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._pending_counts = (host, event, B, st["H"])
        groups = (counts.to(torch.float64) / max(B, 1)).to(torch.float32)
        return [groups[i] for i in range(self.n_bits)], [levels[i] for i in range(self.n_bits)]

    def forward(self, latent):
        with torch.no_grad():
            latent = require_device_input(latent, "latent")
            if latent.dtype != torch.float32:
                latent = latent.float()
            st = self.packed()
            if st["index"] is not None:
                index = st["index"]
                padded = torch.zeros((latent.shape[0], st["H"]), device=latent.device)
                padded[:, index >= 0] = latent[:, index[index >= 0]]
                latent = padded
            return self.decode_bits(ops.pack_bits_gt(latent, 0.5))

    def apply_secant_grad(self):
        raise NotImplementedError(_TRAINING_ONLY)


class QuantizedMatryoshkaSAE(ops.GraphForwardMixin, SparseAutoencoder):
    """``forward(x) -> (latent_groups, reconstruction_levels)``; ``top_k`` is stored and unused, as
    in the reference (sae/quantized_matryoshka.py:192-220)."""

    def __init__(self, input_dim, hidden_dim, top_k, abs_range=4, n_bits=8, allow_bias=True):
        super().__init__(input_dim, hidden_dim)
        self.n_bits = n_bits
        self.abs_range = abs_range
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.allow_bias = allow_bias
        self.top_k = top_k
        lin = nn.Linear(input_dim, hidden_dim)
        nn.init.xavier_uniform_(lin.weight, gain=1)
        nn.init.zeros_(lin.bias)
        self.encoder = HipEncoder(lin, nn.Sigmoid())
        self.decoder = QuantizedMatryoshkaDecoder(hidden_dim, input_dim, abs_range=abs_range, n_bits=n_bits,
                                                  top_k=self.top_k, allow_bias=self.allow_bias)
        self._enc_cache = PackedCache()
        ops.module_handle(self)

    def _encoder_params(self):
        """Encoder weight/bias in the decoder's packed hidden order (inert pad units get a zero row
        and bias -1, so their z bit is 0)."""
        lin = self.encoder.linear
        if not self.decoder.needs_padding:
            return lin.weight.detach(), lin.bias.detach()

        def build():
            index = self.decoder.padded_index(lin.weight.device)
            valid = index >= 0
            W = torch.zeros((index.numel(), self.input_dim), device=lin.weight.device)
            b = -torch.ones((index.numel(),), device=lin.weight.device)
            W[valid] = lin.weight.detach()[index[valid]]
            b[valid] = lin.bias.detach()[index[valid]]
            return {"W": W, "b": b}
        st = self._enc_cache.get((lin.weight, lin.bias), build)
        return st["W"], st["b"]

    #: "auto" | "dense" | "prefilter" | "band" -- how the z bits are computed; the bits are identical on every path.
    #: dense: every latent from the exact-fp32 MFMA contraction.  prefilter: the fp16 candidate sweep lists the units near or
    #: above the sigmoid cutoff, latents inside the error band are re-evaluated exactly; the decoder walks the active units --
    #: pays off when few units fire per row.  band: an fp16 MFMA pass classifies EVERY latent and only the ~1 % inside the
    #: band are re-evaluated (qsae_encode_bits_band) -- for dense activations, where the lists of the prefilter overflow.
    #: auto: prefilter for large batches until a batch shows dense activations (more than half of its rows overflow their
    #: candidate lists, i.e. more than ~8 % of the units fire), then band (dense where the shape is not covered) for this model.
    bits_path = "auto"
    _PREFILTER_MIN_ROWS = 2048

    def resolved_bits_path(self, batch_rows: int) -> str:
        path = self.bits_path
        if path not in ("auto", "dense", "prefilter", "band"):
            raise ValueError(f"bits_path must be 'auto', 'dense', 'prefilter' or 'band', got {path!r}")
        W, _ = self._encoder_params()
        ok = ops.encode_bits_prefilter_supported(batch_rows, self.input_dim, W.shape[0])
        ok_band = ops.encode_bits_band_supported(batch_rows, self.input_dim, W.shape[0])
        if path == "auto":
            big = batch_rows >= self._PREFILTER_MIN_ROWS and W.shape[0] >= 2048
            if not big:
                path = "dense"
            elif getattr(self, "_dense_regime", False):
                path = "band" if ok_band else "dense"
            else:
                path = "prefilter" if ok else ("band" if ok_band else "dense")
        if path == "prefilter" and not ok:
            path = "dense"
        if path == "band" and not ok_band:
            path = "dense"
        return path

    def _prefilter_weights(self):
        W, b = self._encoder_params()
        if not hasattr(self, "_pref_cache"):
            self._pref_cache = PackedCache()
        lin = self.encoder.linear

        def build():
            self._dense_regime = False                     # new weights: probe the activation density again
            Wq, meta = ops.prefilter_pack_w(W, b)
            return {"Wq": Wq, "meta": meta}
        return self._pref_cache.get((lin.weight, lin.bias), build)

    def activation_bits(self, x, path: str = None) -> torch.Tensor:
        """int32-packed z = (sigmoid(encoder pre-activation) > 0.5) in packed hidden order."""
        with torch.no_grad():
            W, b = self._encoder_params()
            x = require_device_input(x, "x")
            path = path or self.resolved_bits_path(x.shape[0])
            if path == "prefilter":
                pw = self._prefilter_weights()
                z, flagged = ops.encode_bits_prefilter(x.float(), W, b, pw["Wq"], pw["meta"])
                self.last_flagged_rows = flagged
                if flagged * 2 > x.shape[0]:                  # the exact fallback of half the rows costs what the dense kernel does
                    self._dense_regime = True
                return z
            if path == "band":
                pw = self._prefilter_weights()
                z, self.last_flagged_rows = ops.encode_bits_band(x.float(), W, b, pw["Wq"], pw["meta"])
                return z
            return ops.encode_bits(x, W, b)

    def forward(self, x):
        if torch.compiler.is_compiling():                  # one graph node: torch.ops.qsae.levels_sae_forward
            lin, dec = self.encoder.linear, self.decoder
            with torch.no_grad():
                groups, levels = torch.ops.qsae.levels_sae_forward(
                    x, [lin.weight, lin.bias, dec.weight, dec.weight_mirror, dec.bias], self._qsae_handle)
            return [groups[i] for i in range(self.n_bits)], [levels[i] for i in range(self.n_bits)]
        return self._forward_eager(x)

    def _forward_eager(self, x):
        with torch.no_grad():
            x = require_device_input(x, "x")
            self.decoder.active_fraction_hint()            # reads the previous call's counts before anything is queued
            path = self.resolved_bits_path(x.shape[0])
            z = self.activation_bits(x, path)
            # few flagged rows = few active units: walk them; otherwise the decoder decides from the last batch's density
            sparse = True if (path == "prefilter" and self.last_flagged_rows * 8 <= x.shape[0]) else None
            return self.decoder.decode_bits(z, sparse=sparse)

    def forward_submit(self, x, slot: int = 0):
        """Queue one forward without waiting for the GPU (see BinarySAE.forward_submit): the z bits of the fp16 candidate
        sweep are queued here (qsae_encode_bits_prefilter_submit); ``result()`` takes the count of rows that need the
        exact dense kernel, queues those and the decoder, and returns ``(latent_groups, reconstruction_levels)``.
        Batches in flight together need different ``slot`` numbers; models on the dense path compute eagerly."""
        with torch.no_grad():
            xd = require_device_input(x, "x")
            path = self.resolved_bits_path(xd.shape[0])
            if path not in ("prefilter", "band"):
                return _SubmittedMatryoshka(self, None, self.forward(xd), xd.shape[0], path)
            W, b = self._encoder_params()
            pw = self._prefilter_weights()
            pending = ops.encode_bits_prefilter_submit(xd.float(), W, b, pw["Wq"], pw["meta"], slot=slot, band=(path == "band"), owner=self._qsae_handle)
            return _SubmittedMatryoshka(self, pending, None, xd.shape[0], path)


class _SubmittedMatryoshka:
    def __init__(self, model, pending, outs, rows, path):
        self._model, self._pending, self._outs, self._rows, self._path = model, pending, outs, rows, path

    def result(self):
        with torch.no_grad():
            if self._pending is not None:
                m = self._model
                z = self._pending.finish()
                flagged = m.last_flagged_rows = self._pending.flagged_rows
                self._pending = None
                if self._path == "prefilter" and flagged * 2 > self._rows:
                    m._dense_regime = True
                m.decoder.active_fraction_hint()
                sparse = True if (self._path == "prefilter" and flagged * 8 <= self._rows) else None
                self._outs = m.decoder.decode_bits(z, sparse=sparse)
            return self._outs
