#!/usr/bin/env python3
"""Small batches (serving-sized): BinarySAE.forward latency per call, eager launches against a captured HIP graph
(torch.cuda.CUDAGraph).  The in-place path (dense fp32 MFMA encoder -> in-place top-k -> sparse decode) has no host
read-back, so the whole forward is capturable; the candidate-sweep path of large batches is not (its flagged-row count
returns to the host) and is not launch-bound either."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
model.decoder.packed()
for B in (1, 16, 64, 256, 1024):
    x = torch.randn((B, 512), device=dev)
    for _ in range(3):
        out = model(x)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        out = model(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        gout = model(x)
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(gout[:2], out[:2]))
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n * 1e6
    print(f"B = {B:5d}: eager {eager:7.1f} us per forward, captured graph {graph:7.1f} us, same bits {same}")
