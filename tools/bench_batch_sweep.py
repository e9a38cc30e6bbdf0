#!/usr/bin/env python3
"""BinarySAE(512, 32768, n_bits=4) forward() wall time vs batch size (default path), one GPU."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE  # noqa: E402

dev = "cuda:0"
D, H = 512, 32768
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
with torch.no_grad():
    model.decoder.weight.copy_(torch.where(torch.rand(model.decoder.weight.shape, device=dev, generator=g) > 0.5, 30.0, -30.0))
    for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        x = torch.randn((B, D), device=dev, generator=g)
        for _ in range(3):
            out = model(x)
        torch.cuda.synchronize()
        iters = 10
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        print(json.dumps(dict(rows=B, path=model.resolved_latent_path(B), ms_per_batch=round(ms, 3),
                              rows_per_s=round(B / ms * 1e3))), flush=True)
        del out, x
