#!/usr/bin/env python3
"""z bits of a dense-regime encoder (zero bias: half of the units fire), 65536 x 512 -> 32768: the exact fp32 contraction
(qsae_encode_bits) against the fp16 classification + band resolution (qsae_encode_bits_band)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import ops  # noqa: E402

dev = "cuda:0"
B, D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 512, 32768
g = torch.Generator(device=dev); g.manual_seed(0)
W = (torch.rand((H, D), device=dev, generator=g) * 2 - 1) * (6.0 / (D + H)) ** 0.5
x = torch.randn((B, D), device=dev, generator=g)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


for shift in (0.0, -1.0, -2.0):
    b = torch.full((H,), shift * 0.2436, device=dev)
    Wq, meta = ops.prefilter_pack_w(W, b)
    z0 = ops.encode_bits(x, W, b)
    z1, flagged = ops.encode_bits_band(x, W, b, Wq, meta)
    dens = float((z0.view(torch.uint8).unsqueeze(-1) >> torch.arange(8, device=dev, dtype=torch.uint8) & 1).sum()) / (B * H)
    print(f"bias {shift:+.1f} sigma (density {dens:.3f}): equal {torch.equal(z0, z1)}, flagged {flagged}; exact fp32 "
          f"{timeit(lambda: ops.encode_bits(x, W, b)):.2f} ms, band {timeit(lambda: ops.encode_bits_band(x, W, b, Wq, meta)):.2f} ms")
