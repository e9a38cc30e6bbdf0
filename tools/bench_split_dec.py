#!/usr/bin/env python3
"""Dense decoders: the exact-fp32 MFMA kernels against the bf16 split kernels (csrc/split_dec_bf16.h), 65536 x 32768 x 512.
usage: python tools/bench_split_dec.py [rows]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
D, H, n_bits = 512, 32768, 4
g = torch.Generator(device=dev); g.manual_seed(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


w = torch.randn((D, H), device=dev, generator=g) * 0.5
codes = ops.pack_ternary(w)
tq = ops.expand_codes_bf16(codes, D, H)
h = torch.relu(torch.randn((B, H), device=dev, generator=g))
flop = 2.0 * B * H * D
t32 = timeit(lambda: ops.decode_ternary_dense(h, codes, D))
tsp = timeit(lambda: ops.decode_ternary_dense_split(h, tq, D))
print(f"ternary decode {B} x {H} x {D}: fp32 MFMA {t32:.2f} ms ({flop / t32 / 1e9:.0f} TF), bf16 split {tsp:.2f} ms "
      f"({3 * flop / tsp / 1e9:.0f} TF of bf16 MFMA = {3 * flop / tsp / 1e9 / 2500:.2f} of 2.5 PF; activations {4.0 * B * H / tsp / 1e6:.0f} GB/s)")
del h
wd, wm = torch.rand((H, D), device=dev, generator=g) * 2 - 1, torch.rand((H, D), device=dev, generator=g) * 2 - 1
mc, ms = ops.pack_matryoshka(wd, wm, n_bits, 4.0)
mtq, s3 = ops.expand_codes_bf16(mc, D, H), ops.split_scale_bf16(ms)
z = torch.randint(-2**31, 2**31 - 1, (B, H // 32), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
bias = torch.randn((D,), device=dev, generator=g)
t32 = timeit(lambda: ops.decode_matryoshka(z, H, D, n_bits, mc, ms, bias, True))
tsp = timeit(lambda: ops.decode_matryoshka_split(z, H, D, n_bits, mtq, s3, bias, True))
print(f"matryoshka decode (4 levels, half of the units active): fp32 MFMA {t32:.2f} ms, bf16 split {tsp:.2f} ms "
      f"({3 * flop / tsp / 1e9:.0f} TF of bf16 MFMA = {3 * flop / tsp / 1e9 / 2500:.2f} of 2.5 PF)")
