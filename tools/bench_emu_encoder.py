#!/usr/bin/env python3
"""Dense ReLU encoder 65536 x 512 -> 32768: the exact fp32 MFMA chain against the fp32-accurate emulation on the fp16 pipe."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import ops  # noqa: E402

dev = "cuda:0"
B, D, H = 65536, 512, 32768
g = torch.Generator(device=dev); g.manual_seed(0)
W = (torch.rand((H, D), device=dev, generator=g) * 2 - 1) * (6.0 / (D + H)) ** 0.5
b = torch.randn((H,), device=dev, generator=g) * 0.05
x = torch.randn((B, D), device=dev, generator=g)
Wc, meta2 = ops.emu_pack_w(W)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, e in evs:
        a.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(e) for a, e in evs)
    return t[len(t) // 2]


t32 = timeit(lambda: ops.encode_dense(x, W, b, ops.ACT_RELU))
temu = timeit(lambda: ops.encode_dense_emu(x, Wc, meta2, b, ops.ACT_RELU))
h0, h1 = ops.encode_dense(x[:4096], W, b, ops.ACT_RELU), ops.encode_dense_emu(x[:4096], Wc, meta2, b, ops.ACT_RELU)
print(f"encoder + ReLU {B} x {D} -> {H}: exact fp32 chain {t32:.2f} ms, emulated {temu:.2f} ms "
      f"({3 * 2.0 * B * D * H / temu / 1e9:.0f} TF of fp16 MFMA); max |diff| {float((h0 - h1).abs().max()):.2e} at latent std {float(h0.std()):.3f}")
