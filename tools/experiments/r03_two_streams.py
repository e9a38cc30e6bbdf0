#!/usr/bin/env python3
"""Item 4(a) of the round-2 verdict, the cheap form: two independent submit / finish pipelines on two HIP streams (even
batches on stream A, odd batches on stream B), so that the hardware scheduler is free to start workgroups of one batch's
refinement while the other batch's candidate sweep drains (and vice versa).  Prints ms per step for 1 and 2 streams."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from quantizedsae_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
g = torch.Generator(device=dev); g.manual_seed(7)
xs = [torch.randn((65536, 512), device=dev, generator=g) for _ in range(4)]
model.decoder.packed()


def run(n_streams, steps=40):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
    sq = [torch.zeros((), dtype=torch.float64, device=dev) for _ in streams]
    pend = [None] * len(streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % len(streams)
        with torch.cuda.stream(streams[s]):
            h = model.forward_submit(xs[i % 4], slot=(i // len(streams)) % 2)
            if pend[s] is not None:
                _l, rec, _p = pend[s][0].result()
                ops.sq_err_sum(rec, pend[s][1], sq[s])
            pend[s] = (h, xs[i % 4])
    for s, p in enumerate(pend):
        if p is not None:
            with torch.cuda.stream(streams[s]):
                _l, rec, _p = p[0].result()
                ops.sq_err_sum(rec, p[1], sq[s])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for n in (1, 2, 1, 2, 3):
    run(n, 6)
    print(f"{n} stream(s): {run(n):.3f} ms per 65536-row step")
