#!/usr/bin/env python3
"""Config 5 of BASELINE.json in miniature: a hidden-state chunk file in the reference's on-disk format
([num_contexts, tokens_per_context, 512], data/dataset.py:7-33; synthetic N(0,1) stand-in for the pythia-70m layer-3
residuals, which do not exist offline) streamed through BinarySAE(512, 32768, n_bits=4) in 65536-row device batches,
row-sharded per rank, recon-MSE accumulated on the device.  Prints rows/s with the host->device copies included."""
import json
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, data, ops  # noqa: E402

dev = "cuda:0"
contexts, tokens, D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1600, 250, 512, 32768
torch.manual_seed(0)                                   # the model constructor draws the encoder from the global RNG
g = torch.Generator(); g.manual_seed(100)
with tempfile.TemporaryDirectory() as tmp:
    path = Path(tmp) / "chunk0.pt"
    torch.save(torch.randn((contexts, tokens, D), generator=g).to(torch.float16), path)
    ds = data.HiddenStatesTorchDataset(path)
    model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
    gd = torch.Generator(device=dev); gd.manual_seed(1)
    with torch.no_grad():
        model.decoder.weight.copy_(torch.where(torch.rand(model.decoder.weight.shape, device=dev, generator=gd) > 0.5, 30.0, -30.0))
        for warm in (True, False):
            sq = torch.zeros((), dtype=torch.float64, device=dev)
            rows = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for xb in ds.iter_batches(65536, dev):
                _idx, _val, recon = model.forward_compact(xb)
                ops.sq_err_sum(recon, xb, sq)
                rows += xb.shape[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    print(json.dumps(dict(chunk_shape=[contexts, tokens, D], rows=rows, seconds=round(dt, 4), rows_per_s=round(rows / dt),
                          recon_mse=float(sq.item()) / (rows * D),
                          note="compact outputs (idx, val, reconstruction); fp16 chunk rows copied host -> device and widened there, copies inside the timed loop")))
