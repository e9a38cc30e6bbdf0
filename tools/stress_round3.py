#!/usr/bin/env python3
"""Soak of the round-3 paths on fresh random batches (different data scales and biases every iteration):
  * BinarySAE soft (table decode inside the refinement), BinarySAE polarised (the headline: sliced refinement, 13-instruction
    nibble decode) and BaselineSparseAutoencoder through forward_submit, two in flight, against the exact-fp32 fused path +
    separate decode (bit for bit);
  * qsae_encode_bits_band against qsae_encode_bits (bit for bit) at a random activation density;
  * split ternary / matryoshka decoders against the fp32 kernels (1e-5 relative).
usage: python tools/stress_round3.py [iterations]"""
import sys
import warnings
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BaselineSparseAutoencoder, BinarySAE, ops  # noqa: E402

warnings.simplefilter("ignore")
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, D, H = 65536, 512, 32768
g = torch.Generator(device=dev); g.manual_seed(11)
soft = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
hard = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
base = BaselineSparseAutoencoder(D, H).to(dev).eval()
with torch.no_grad():
    hard.decoder.weight.copy_(torch.where(torch.rand(hard.decoder.weight.shape, device=dev, generator=g) > 0.5, 30.0, -30.0))
    hard.encoder[0].bias.copy_(torch.randn((H,), device=dev, generator=g) * 0.05)
    soft.decoder.weight.copy_(torch.randn(soft.decoder.weight.shape, device=dev, generator=g) * 2.0)
    soft.encoder[0].bias.copy_(torch.randn((H,), device=dev, generator=g) * 0.05)
w = torch.randn((D, H), device=dev, generator=g) * 0.5
codes = ops.pack_ternary(w)
tq = ops.expand_codes_bf16(codes, D, H)
bad = 0
bits = lambda t: t.contiguous().view(torch.int32)
pend = {}
for it in range(N):
    scale = float(10.0 ** (torch.rand((), generator=g, device=dev) * 4 - 2))
    x = torch.randn((B, D), device=dev, generator=g) * scale
    x[torch.randint(0, B, (3,), device=dev, generator=g)] = float("nan") if it % 5 == 0 else 0.0
    for name, model in (("soft", soft), ("hard", hard), ("base", base)):
        h = model.forward_submit(x, slot=it % 2)
        if name in pend:
            px, ph = pend[name]
            got = ph.result()
            model.latent_path = "fused"
            if hasattr(model, "fuse_decode"):
                model.fuse_decode = False
            want = model(px)
            model.latent_path = "auto"
            if hasattr(model, "fuse_decode"):
                model.fuse_decode = True
            ok = all(torch.equal(bits(a), bits(b)) for a, b in zip(got[:2], want[:2]))
            bad += 0 if ok else 1
            if not ok:
                print(f"iteration {it}: {name} MISMATCH", flush=True)
        pend[name] = (x, h)
    # bits: band classification against the exact kernel at a random density
    lin = base.encoder.linear
    shift = float(torch.rand((), generator=g, device=dev) * 3 - 2.5) * 0.18 * scale
    b = torch.full((H,), shift, device=dev)
    Wq, meta = ops.prefilter_pack_w(lin.weight.detach(), b)
    z, flagged = ops.encode_bits_band(x[:16384], lin.weight.detach(), b, Wq, meta)
    if not torch.equal(z, ops.encode_bits(x[:16384], lin.weight.detach(), b)):
        bad += 1
        print(f"iteration {it}: band bits MISMATCH (flagged {flagged})", flush=True)
    # split ternary decoder against the fp32 kernel
    hlat = torch.relu(torch.randn((8192, H), device=dev, generator=g)) * scale
    r0, r1 = ops.decode_ternary_dense(hlat, codes, D), ops.decode_ternary_dense_split(hlat, tq, D)
    rel = float((r0 - r1).abs().max() / r0.abs().max().clamp_min(1e-30))
    if not rel < 1e-5:
        bad += 1
        print(f"iteration {it}: split ternary rel {rel:.2e}", flush=True)
    if it % 5 == 0:
        print(f"iteration {it}: ok so far ({bad} bad), scale {scale:.3g}, band flagged {flagged}, split rel {rel:.1e}", flush=True)
print(f"{N} iterations: {bad} mismatches")
sys.exit(1 if bad else 0)
