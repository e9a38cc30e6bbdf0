#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (configs 3, 4 + baseline/rq) at full size on one GPU:
forward() wall time per 65536-row batch.  Synthetic parameters as in SURVEY.md section 8d."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,  # noqa: E402
                              TernarySparseAutoencoder)

dev = "cuda:0"
D, H = 512, 32768
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.randn((B, D), device=dev, generator=g)


def bench(name, model, flops_per_row, iters=3):
    model = model.to(dev).eval()
    for _ in range(2):
        out = model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    del out
    print(json.dumps(dict(variant=name, rows=B, ms_per_batch=round(ms, 3), rows_per_s=round(B / ms * 1e3),
                          algorithmic_tflops=round(flops_per_row * B / ms / 1e9, 1))), flush=True)
    del model
    torch.cuda.empty_cache()


with torch.no_grad():
    m = TernarySparseAutoencoder(D, H)
    m.decoder.weight.normal_(0, 0.5)
    bench("ternary (config 3)", m, 4.0 * D * H)
    m = QuantizedMatryoshkaSAE(D, H, top_k=32, abs_range=4, n_bits=4)
    m.encoder[0].bias.fill_(-0.44)
    m.decoder.weight.uniform_(-1, 1); m.decoder.weight_mirror.uniform_(-1, 1)
    bench("matryoshka n_bits=4 (config 4)", m, 4.0 * D * H)
    m = BaselineSparseAutoencoder(D, H)
    bench("baseline top-32", m, 2.0 * D * H + 2.0 * 32 * D)
    m = BinarySAE(D, H, gamma=1.5, n_bits=4)
    m.decoder.weight.copy_(torch.where(torch.rand_like(m.decoder.weight) > 0.5, 30.0, -30.0))
    bench("binary gamma=1.5 (registry)", m, 2.0 * D * H + 2.0 * 65 * D)
    if B <= 32768:
        m = ResidualQuantizedSAE(D, H, top_k=32, abs_range=1.5, n_bits=4)
        bench("residual (rq_sae)", m, 4.0 * D * H)
