#!/usr/bin/env python3
"""Per-step wall time of the headline forward over many steps (each step synchronised): looks for outliers, e.g. a
sweep workgroup kept off its CU by early fill waves."""
import json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, ops
dev = "cuda:0"
D, H, B = 512, 32768, 65536
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
x = torch.randn((B, D), device=dev)
acc = torch.zeros((), dtype=torch.float64, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ts = []
with torch.no_grad():
    for i in range(n + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
        torch.cuda.synchronize()
        if i >= 5:
            ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(json.dumps(dict(steps=n, min=round(ts[0], 3), p50=round(ts[n // 2], 3), p90=round(ts[int(n * 0.9)], 3),
                      p99=round(ts[int(n * 0.99)], 3), max=round(ts[-1], 3))))
