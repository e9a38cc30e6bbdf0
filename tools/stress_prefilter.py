#!/usr/bin/env python3
"""Repeatability / equivalence stress of the default path at the headline shape: N runs of encode_topk_prefilter
(with the dense latent) must all return the bits of the exact-fp32 fused path."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import ops  # noqa: E402

B, D, H, k = 65536, 512, 32768, 65
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator(device="cuda:0"); g.manual_seed(7)
W = (torch.rand((H, D), device="cuda:0", generator=g) * 2 - 1) * (6.0 / (D + H)) ** 0.5
bias = torch.randn((H,), device="cuda:0", generator=g) * 0.02
Wq, meta = ops.prefilter_pack_w(W, bias)
bad = 0
for it in range(N):
    x = torch.randn((B, D), device="cuda:0", generator=g)
    ridx, rval, rdense = ops.encode_topk_latent(x, W, bias, k)          # exact fp32 sweep
    idx, val, dense = ops.encode_topk_prefilter(x, W, bias, Wq, meta, k)
    ok = bool(torch.equal(idx, ridx)) and bool(torch.equal(val.view(torch.int32), rval.view(torch.int32))) \
        and bool(torch.equal(dense.view(torch.int32), rdense.view(torch.int32)))
    bad += 0 if ok else 1
    if not ok:
        rows = (idx != ridx).any(dim=1).nonzero().flatten()[:8].tolist()
        print(f"iteration {it}: MISMATCH rows {rows}", flush=True)
    elif it % 10 == 0:
        print(f"iteration {it}: identical", flush=True)
    del ridx, rval, rdense, idx, val, dense
print(f"{N} batches of {B} rows: {bad} mismatching batches")
sys.exit(1 if bad else 0)
