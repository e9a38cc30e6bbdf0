"""Consumers of the sparse latent: activation masks, L0 per level, activation / co-activation counts.

Mirror of the reference's analysis helpers (scripts/analysis/dynamic_analysis.py): same function names,
arguments and result keys, so an analysis script switches over by changing the import.  The reference reduces
the dense [B, H] latent to a boolean mask and forms ``mask.sum(0)`` and ``mask.T @ mask`` (a dense [H,B]x[B,H]
product per batch); here the top-k variants work from the compact ``(idx, val)`` output of
``forward_compact`` -- k and k^2 integer increments per row (``qsae_activation_counts``,
``qsae_coactivation_sparse``) -- and the threshold variants from the bit-packed encoder output.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional

import torch

from .. import torch_ops as ops          # torch.ops.qsae.* (dispatcher ops over the C ABI)
from ..sae import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE)
from .framework import SAEWrapper, _ensure_tensor, compute_reconstruction_error  # noqa: F401  (re-export)


def _hidden_dim(sae: SAEWrapper) -> int:
    """Total number of hidden units (dynamic_analysis.py:17-27)."""
    model = sae.model
    if isinstance(model, (BinarySAE, QuantizedMatryoshkaSAE)):
        return int(model.hidden_dim)
    if isinstance(model, BaselineSparseAutoencoder):
        return int(model.encoder.linear.weight.shape[0])
    if isinstance(model, ResidualQuantizedSAE):
        return int(sum(s.hidden_dim for s in model.saes))
    raise ValueError(f"Unable to determine hidden_dim for model type {type(model)}")


def _bits_to_mask(zbits: torch.Tensor, index: Optional[torch.Tensor], H: int) -> torch.Tensor:
    """int32 [B, words] packed bits (bit j of word w = packed unit 32 w + j) -> bool [B, H] in the model's own
    hidden order (``index[p]`` = original unit of packed position p, -1 for padding)."""
    B, words = zbits.shape
    shifts = torch.arange(32, device=zbits.device, dtype=torch.int32)
    bits = ((zbits.unsqueeze(-1) >> shifts) & 1).to(torch.bool).reshape(B, words * 32)
    if index is None:
        return bits[:, :H]
    mask = torch.zeros((B, H), dtype=torch.bool, device=zbits.device)
    valid = index >= 0
    mask[:, index[valid]] = bits[:, valid]
    return mask


def _stage_bits(model: QuantizedMatryoshkaSAE, x: torch.Tensor):
    """(packed bits, packed-position -> original-unit index or None) of one matryoshka encoder."""
    zb = model.activation_bits(x)
    index = model.decoder.padded_index(zb.device) if model.decoder.needs_padding else None
    return zb, index


def _residual_stages(model: ResidualQuantizedSAE, x: torch.Tensor):
    """Yields (stage, packed bits, index) with the residual updated as in the forward pass
    (sae/residual_quantized.py:53-69; dynamic_analysis.py:56-70)."""
    residual = x if x.dtype == torch.float32 else x.float()
    for sub in model.saes:
        zb, index = _stage_bits(sub, residual)
        _, levels = sub.decoder.decode_bits(zb)
        yield sub, zb, index
        residual = (residual - levels[-1]) * 2


def activation_indices(sae: SAEWrapper, x: torch.Tensor):
    """Compact activation set of the top-k variants: (idx int32 [B,k], val fp32 [B,k]); an entry is active
    when val > 0 (the reference's ``latent > 0``)."""
    model = sae.model
    if not isinstance(model, (BinarySAE, BaselineSparseAutoencoder)):
        raise TypeError(f"{type(model).__name__} has no compact top-k output; use _activation_mask")
    idx, val, _recon = model.forward_compact(x)
    return idx, val


def _activation_mask(sae: SAEWrapper, x: torch.Tensor) -> torch.Tensor:
    """Boolean mask [batch, hidden_dim] of the active features, on the CPU like the reference's
    (dynamic_analysis.py:30-73): BinarySAE / baseline ``latent > 0``, matryoshka ``sigmoid(encoder) > 0.5``,
    residual: the stages' masks concatenated."""
    model = sae.model
    H = _hidden_dim(sae)
    with torch.no_grad():
        x = _ensure_tensor(x).to(sae.device)
        if isinstance(model, (BinarySAE, BaselineSparseAutoencoder)):
            idx, val = activation_indices(sae, x)
            mask = torch.zeros((x.shape[0], H), dtype=torch.bool, device=x.device)
            mask.scatter_(1, idx.long(), val > 0)
        elif isinstance(model, QuantizedMatryoshkaSAE):
            zb, index = _stage_bits(model, x)
            mask = _bits_to_mask(zb, index, H)
        elif isinstance(model, ResidualQuantizedSAE):
            parts = [_bits_to_mask(zb, index, sub.hidden_dim) for sub, zb, index in _residual_stages(model, x)]
            mask = torch.cat(parts, dim=1)
        else:
            raise TypeError(f"Unsupported SAE model type: {type(model)}")
    return mask.cpu()


def _packed_counts_to_units(counts_packed: torch.Tensor, index: Optional[torch.Tensor], H: int) -> torch.Tensor:
    if index is None:
        return counts_packed[:H]
    out = torch.zeros((H,), dtype=torch.int64, device=counts_packed.device)
    valid = index >= 0
    out[index[valid]] = counts_packed[valid]
    return out


def compute_l0_by_level(sae: SAEWrapper, loader: Iterable[Any], device: Optional[Any] = None) -> torch.Tensor:
    """Average number of active units per token for each level (dynamic_analysis.py:168-251): matryoshka
    levels = nested dictionary slices, residual levels = stages, other SAEs a length-1 tensor."""
    if device is not None:
        sae.to(device)
    sae.eval()
    model = sae.model
    n_tokens = 0
    with torch.no_grad():
        if isinstance(model, QuantizedMatryoshkaSAE):
            sizes = list(model.decoder.nested_dictionary_size)
            H = int(model.hidden_dim)
            per_unit = torch.zeros((H,), dtype=torch.int64, device=sae.device)
            for batch in loader:
                x = _ensure_tensor(batch).to(sae.device)
                zb, index = _stage_bits(model, x)
                per_unit += _packed_counts_to_units(ops.activation_counts_bits(zb), index, H)
                n_tokens += x.shape[0]
            bounds = torch.tensor([0] + sizes).cumsum(0).tolist()
            total = torch.stack([per_unit[bounds[i]:bounds[i + 1]].sum() for i in range(len(sizes))])
            return total.to(torch.float64).cpu() / max(float(n_tokens), 1.0)
        if isinstance(model, ResidualQuantizedSAE):
            total = torch.zeros((len(model.saes),), dtype=torch.float64)
            for batch in loader:
                x = _ensure_tensor(batch).to(sae.device)
                for i, (sub, zb, index) in enumerate(_residual_stages(model, x)):
                    total[i] += float(ops.activation_counts_bits(zb).sum().item())   # padding bits are never set
                n_tokens += x.shape[0]
            return total / max(float(n_tokens), 1.0)
        total_act = 0.0
        for batch in loader:
            x = _ensure_tensor(batch).to(sae.device)
            _idx, val = activation_indices(sae, x)
            total_act += float((val > 0).sum().item())
            n_tokens += x.shape[0]
        return torch.tensor([total_act / max(float(n_tokens), 1.0)], dtype=torch.float64)


def _tokens_per_feature(feat: torch.Tensor, tok: torch.Tensor, H: int, into: List[List[int]]) -> None:
    """Append token ids per feature; within a feature in ascending row order, like the reference's loop over
    ``mask.nonzero()`` (row-major)."""
    if feat.numel() == 0:
        return
    order = torch.sort(feat, stable=True).indices
    feat_s, tok_s = feat[order].cpu(), tok[order].cpu()
    uniq, counts = torch.unique_consecutive(feat_s, return_counts=True)
    start = 0
    toks = tok_s.tolist()
    for f, c in zip(uniq.tolist(), counts.tolist()):
        into[f].extend(toks[start:start + c])
        start += c


def compute_activation_stats(sae: SAEWrapper, loader: Iterable[Any], *, token_ids: torch.Tensor,
                             tokens_per_context: int, device: Optional[Any] = None,
                             with_tokens: bool = True) -> Dict[str, Any]:
    """activation_counts [H] (int64), coactivation [H,H] (int32, mask^T mask) and tokens_per_feature
    (dynamic_analysis.py:255-311).  Counts are accumulated on the device and copied to the host once."""
    if device is not None:
        sae.to(device)
    sae.eval()
    model = sae.model
    H = _hidden_dim(sae)
    dev = sae.device
    counts = torch.zeros((H,), dtype=torch.int64, device=dev)
    coact = torch.zeros((H, H), dtype=torch.int32, device=dev)
    tokens_per_feature: List[List[int]] = [[] for _ in range(H)]
    global_index = 0
    compact = isinstance(model, (BinarySAE, BaselineSparseAutoencoder))
    with torch.no_grad():
        for batch in loader:
            x = _ensure_tensor(batch).to(dev)
            B = x.shape[0]
            flat = torch.arange(global_index, global_index + B, dtype=torch.long)
            batch_tok = token_ids[torch.div(flat, tokens_per_context, rounding_mode="floor"), flat % tokens_per_context]
            if compact:
                idx, val = activation_indices(sae, x)
                ops.activation_counts(idx, val, H, counts)
                ops.coactivation_sparse(idx, val, H, coact)
                if with_tokens:
                    on = val > 0
                    rows = torch.arange(B, device=dev).unsqueeze(1).expand_as(idx)[on]
                    _tokens_per_feature(idx[on].long(), batch_tok.to(dev)[rows], H, tokens_per_feature)
            else:
                # threshold variants: hundreds to thousands of active units per row, so the mask product is formed
                # densely like the reference's analyze_dataset does (dynamic_analysis.py:405-415) -- with the exact-fp32
                # MFMA contraction of this package (qsae_encode_dense)
                mask = _activation_mask(sae, x).to(dev)
                counts += mask.sum(dim=0)
                pad = (-mask.shape[0]) % 4                               # the contraction wants K % 4 == 0: zero rows add nothing
                mt = torch.nn.functional.pad(mask.t().float(), (0, pad)).contiguous()   # [H, B(+pad)]: both GEMM operands
                coact += ops.encode_dense(mt, mt, None, ops.ACT_NONE).to(torch.int32)   # exact: 0/1 products, sums < 2^24
                if with_tokens:
                    nz = mask.nonzero(as_tuple=False)
                    _tokens_per_feature(nz[:, 1], batch_tok.to(dev)[nz[:, 0]], H, tokens_per_feature)
            global_index += B
    return {"activation_counts": counts.cpu(), "coactivation": coact.cpu(), "tokens_per_feature": tokens_per_feature}


def compute_reconstruction_error_by_level(sae: SAEWrapper, loader: Iterable[Any], device: Optional[Any] = None) -> torch.Tensor:
    """Per-level reconstruction MSE (dynamic_analysis.py:103-165): matryoshka = every cumulative level against the
    input; residual = every stage's reconstruction against the residual it was given (the training objective);
    other SAEs a length-1 tensor with the overall MSE.  Squared errors are summed on the device in fp64
    (qsae_sq_err_sum) and read once at the end."""
    if device is not None:
        sae.to(device)
    sae.eval()
    model = sae.model
    if not isinstance(model, (QuantizedMatryoshkaSAE, ResidualQuantizedSAE)):
        return torch.tensor([compute_reconstruction_error(sae, loader)], dtype=torch.float64)
    sums: Optional[List[torch.Tensor]] = None
    n_elements = 0
    with torch.no_grad():
        for batch in loader:
            x = _ensure_tensor(batch).to(sae.device)
            x = x if x.dtype == torch.float32 else x.float()
            _groups, levels = model(x)
            if sums is None:
                sums = [torch.zeros((), dtype=torch.float64, device=sae.device) for _ in levels]
            target = x.contiguous()
            for i, recon in enumerate(levels):
                ops.sq_err_sum(recon, target, sums[i])
                if isinstance(model, ResidualQuantizedSAE):
                    target = ((target - recon) * 2).contiguous()
            n_elements += x.numel()
    if sums is None:
        raise ValueError("empty loader")
    return torch.stack(sums).cpu() / float(max(n_elements, 1))


def analyze_dataset(sae: SAEWrapper, loader: Iterable[Any], *, token_ids: torch.Tensor, tokens_per_context: int,
                    device: Optional[Any] = None, with_tokens: bool = True) -> Dict[str, Any]:
    """One pass over the data: final reconstruction MSE, activation counts, co-activation matrix and tokens per
    feature (dynamic_analysis.py:317-440; same result keys, ``mse_per_level`` / ``l0_per_level`` are None there
    too).  Top-k variants run ``forward_compact`` once per batch and feed its (idx, val, reconstruction) to the
    integer kernels and the fp64 squared-error sum; the threshold variants take the reconstruction from the
    forward pass and the masks from the bit-packed encoder output."""
    if device is not None:
        sae.to(device)
    sae.eval()
    model = sae.model
    H = _hidden_dim(sae)
    dev = sae.device
    counts = torch.zeros((H,), dtype=torch.int64, device=dev)
    coact = torch.zeros((H, H), dtype=torch.int32, device=dev)
    sq = torch.zeros((), dtype=torch.float64, device=dev)
    tokens_per_feature: List[List[int]] = [[] for _ in range(H)]
    global_index, n_elements = 0, 0
    compact = isinstance(model, (BinarySAE, BaselineSparseAutoencoder))
    with torch.no_grad():
        for batch in loader:
            x = _ensure_tensor(batch).to(dev)
            x = (x if x.dtype == torch.float32 else x.float()).contiguous()
            B = x.shape[0]
            flat = torch.arange(global_index, global_index + B, dtype=torch.long)
            batch_tok = token_ids[torch.div(flat, tokens_per_context, rounding_mode="floor"), flat % tokens_per_context]
            if compact:
                idx, val, recon = model.forward_compact(x)
                ops.sq_err_sum(recon, x, sq)
                ops.activation_counts(idx, val, H, counts)
                ops.coactivation_sparse(idx, val, H, coact)
                if with_tokens:
                    on = val > 0
                    rows = torch.arange(B, device=dev).unsqueeze(1).expand_as(idx)[on]
                    _tokens_per_feature(idx[on].long(), batch_tok.to(dev)[rows], H, tokens_per_feature)
            else:
                ops.sq_err_sum(sae(x)["reconstruction"].to(dev).contiguous(), x, sq)
                mask = _activation_mask(sae, x).to(dev)
                counts += mask.sum(dim=0)
                pad = (-mask.shape[0]) % 4
                mt = torch.nn.functional.pad(mask.t().float(), (0, pad)).contiguous()
                coact += ops.encode_dense(mt, mt, None, ops.ACT_NONE).to(torch.int32)
                if with_tokens:
                    nz = mask.nonzero(as_tuple=False)
                    _tokens_per_feature(nz[:, 1], batch_tok.to(dev)[nz[:, 0]], H, tokens_per_feature)
            global_index += B
            n_elements += x.numel()
    return {"mse_final": float(sq.item()) / max(n_elements, 1), "mse_per_level": None, "l0_per_level": None,
            "activation_counts": counts.cpu(), "coactivation": coact.cpu(), "tokens_per_feature": tokens_per_feature}
