#!/usr/bin/env python3
"""Activation-statistics consumers at the headline shape: compact path (forward_compact + qsae_activation_counts +
qsae_coactivation_sparse) against the reference's formulation on the same GPU (dense mask, mask^T @ mask in fp32)."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, ops  # noqa: E402

dev = "cuda:0"
D, H, B = 512, 32768, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
x = torch.randn((B, D), device=dev, generator=g)
counts = torch.zeros((H,), dtype=torch.int64, device=dev)
coact = torch.zeros((H, H), dtype=torch.int32, device=dev)


def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


with torch.no_grad():
    idx, val, _ = model.forward_compact(x)
    t_fwd = timed(lambda: model.forward_compact(x))
    t_cnt = timed(lambda: ops.activation_counts(idx, val, H, counts))
    t_co = timed(lambda: ops.coactivation_sparse(idx, val, H, coact))
    print(json.dumps(dict(rows=B, forward_compact_ms=round(t_fwd, 3), activation_counts_ms=round(t_cnt, 3),
                          coactivation_sparse_ms=round(t_co, 3))), flush=True)
    # the reference's formulation, 8192 rows at a time (a [8192, 32768] fp32 mask is 1 GiB)
    Bd = 8192
    lat, _, _ = model(x[:Bd])
    def dense():
        m = (lat > 0).float()
        return torch.matmul(m.t(), m)
    t_dense = timed(dense, iters=2)
    print(json.dumps(dict(rows=Bd, dense_mask_matmul_ms=round(t_dense, 3),
                          per_65536_rows_ms=round(t_dense * B / Bd, 1))), flush=True)
