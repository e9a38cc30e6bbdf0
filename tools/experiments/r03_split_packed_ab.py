#!/usr/bin/env python3
"""A/B of the two dictionary forms of the split decoders: bf16 image streamed by LDS-DMA (32 KiB per K step) against 2-bit
codes staged as they are (4 KiB per K step) and expanded at fragment-read time through an LDS table.  Same bits expected."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
B, D, H, n_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 512, 32768, 4
g = torch.Generator(device=dev); g.manual_seed(0)
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)


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
tc = torch.empty((H // 32, D), dtype=torch.int64, device=dev)
ops.check(lib.qsae_codes_steps_image(p(codes), D, H, p(tc), st()))
h = torch.relu(torch.randn((B, H), device=dev, generator=g))
r0 = ops.decode_ternary_dense_split(h, tq, D)
r1 = torch.empty_like(r0)
packed = lambda: ops.check(lib.qsae_decode_ternary_dense_split_packed(p(h), H, B, H, p(tc), D, p(r1), st()))
packed()
print("ternary: same bits", torch.equal(r0, r1), f"bf16 image {timeit(lambda: ops.decode_ternary_dense_split(h, tq, D)):.2f} ms, "
      f"packed codes {timeit(packed):.2f} ms")
del h
wd, wm = torch.rand((H, D), device=dev, generator=g) * 2 - 1, torch.rand((H, D), device=dev, generator=g) * 2 - 1
mc, ms = ops.pack_matryoshka(wd, wm, n_bits, 4.0)
mtq, s3 = ops.expand_codes_bf16(mc, D, H), ops.split_scale_bf16(ms)
mtc = torch.empty((H // 32, D), dtype=torch.int64, device=dev)
ops.check(lib.qsae_codes_steps_image(p(mc), D, H, p(mtc), st()))
z = torch.randint(-2**31, 2**31 - 1, (B, H // 32), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
bias = torch.randn((D,), device=dev, generator=g)
l0, c0 = ops.decode_matryoshka_split(z, H, D, n_bits, mtq, s3, bias, True)
l1 = torch.empty_like(l0)
c1 = torch.zeros_like(c0)
mp = lambda: ops.check(lib.qsae_decode_matryoshka_split_packed(p(z), H // 32, B, H, D, n_bits, None, p(mtc), p(s3), p(bias), 1, p(l1), p(c1), st()))
mp()
print("matryoshka: same bits", torch.equal(l0, l1) and torch.equal(c0, c1),
      f"bf16 image {timeit(lambda: ops.decode_matryoshka_split(z, H, D, n_bits, mtq, s3, bias, True)):.2f} ms, packed codes {timeit(mp):.2f} ms")
