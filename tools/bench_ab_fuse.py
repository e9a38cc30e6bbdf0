#!/usr/bin/env python3
"""A/B in one process: BinarySAE.forward with the decode inside the refinement kernel (1) vs a separate launch (0)."""
import json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, ops
dev = "cuda:0"
D, H, B = 512, 32768, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
with torch.no_grad():
    model.decoder.weight.copy_(torch.where(torch.rand_like(model.decoder.weight) > 0.5, 30.0, -30.0))
x = torch.randn((B, D), device=dev, generator=g)
acc = torch.zeros((), dtype=torch.float64, device=dev)
res, ref = {}, None
with torch.no_grad():
    for rnd in range(4):
        for fuse in (False, True):
            model.fuse_decode = fuse
            for _ in range(2):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            res.setdefault(fuse, []).append((time.perf_counter() - t0) / 10 * 1e3)
            if ref is None:
                ref = (lat.clone(), rec.clone())
            else:
                assert torch.equal(ref[0], lat) and torch.equal(ref[1], rec)
for fuse, v in res.items():
    print(json.dumps(dict(fuse_decode=fuse, ms_per_step=[round(t, 3) for t in v], median=round(sorted(v)[len(v) // 2], 3))))
