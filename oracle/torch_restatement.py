"""torch-CPU restatement of the reference's *op sequence* -- TEST INFRASTRUCTURE ONLY.

Where oracle/qsae_oracle.c fixes the arithmetic (it is the bit-exact checker), this file
restates what the reference actually executes per forward, op by op, with stock ATen CPU
kernels: it is the timing counterpart for bench.py's ``cpu_baseline`` (the reference's
Python cannot travel to the GPU box) and is itself checked against the golden vectors in
tests/test_oracle_golden.py::test_torch_restatement_matches_reference.

BinarySAE (sae/binary.py:91-103, 24-47):
    F.linear -> topk -> zeros_like/scatter_/mul -> sigmoid(logits) -> bit-weight reduce ->
    dense matmul -> *step + bias -> polarize mean
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


@torch.no_grad()
def binary_forward(x, enc_w, enc_b, dec_logits, dec_bias, *, n_bits: int, gamma: float, k: int):
    latent = F.linear(x, enc_w, enc_b)
    _, idx = latent.topk(k, dim=1)
    mask = torch.zeros_like(latent)
    mask.scatter_(1, idx, 1.0)
    sparse = latent * mask
    p = torch.sigmoid(dec_logits)
    bw = 2.0 ** torch.arange(n_bits, dtype=p.dtype)
    bw[-1] *= -1
    pr = p.view(dec_logits.shape[0], -1, n_bits)
    int_w = (pr * bw).sum(-1).float()
    recon = (gamma / 2 ** (n_bits - 1)) * sparse.matmul(int_w) + dec_bias
    bwp = 2.0 ** torch.arange(n_bits, dtype=p.dtype)
    polarize = (pr * (1 - pr) * bwp).mean()
    return sparse, recon, polarize


@torch.no_grad()
def baseline_forward(x, enc_w, enc_b, dec_w, dec_b, *, k: int = 32):
    h = F.linear(x, enc_w, enc_b)
    vals, idx = torch.topk(h, k, dim=1)
    hs = torch.zeros_like(h)
    hs.scatter_(1, idx, vals)
    return hs, F.linear(hs, dec_w, dec_b)
