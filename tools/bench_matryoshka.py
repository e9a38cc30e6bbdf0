#!/usr/bin/env python3
"""QuantizedMatryoshkaSAE(512, 32768, n_bits=4) forward, 65536 rows: exact dense kernels vs candidate sweep + sparse walk,
per encoder-bias shift (standard deviations of the latent => fraction of units that fire)."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import QuantizedMatryoshkaSAE, ops  # noqa: E402

B, D, H = 65536, 512, 32768
dev = torch.device("cuda:0")
x = torch.randn(B, D, device=dev)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n, r


shifts = [float(a) for a in sys.argv[1:]] or [-3.0, -2.5, -2.0]
for shift in shifts:
    m = QuantizedMatryoshkaSAE(D, H, top_k=32, abs_range=4, n_bits=4).to(dev).eval()
    with torch.no_grad():
        sigma = (D ** 0.5) * (6.0 / (D + H)) ** 0.5 / 3 ** 0.5
        m.encoder[0].bias.fill_(shift * sigma)
        m.decoder.weight.uniform_(-1, 1)
        m.decoder.weight_mirror.uniform_(-1, 1)
    m.decoder.packed()
    W, b = m._encoder_params()
    pw = m._prefilter_weights()
    st = m.decoder.packed()
    t_bits_dense, z0 = timed(lambda: ops.encode_bits(x, W, b))
    t_bits_pref, (z1, flagged) = timed(lambda: ops.encode_bits_prefilter(x, W, b, pw["Wq"], pw["meta"]))
    assert torch.equal(z0, z1)
    args = (st["H"], D, 4)
    t_dec_dense, (l0, c0) = timed(lambda: ops.decode_matryoshka(z0, *args, st["codes"], st["scale"], m.decoder.bias.detach(), True, st["sizes"]))
    t_dec_sparse, (l1, c1) = timed(lambda: ops.decode_matryoshka_sparse(z0, *args, st["codes_rows"], st["scale"], m.decoder.bias.detach(), True, st["sizes"]))
    assert torch.equal(l0, l1) and torch.equal(c0, c1)
    del l0, l1
    m.bits_path = "dense"
    m.decoder.SPARSE_MAX_ACTIVE_FRACTION = 0.0             # dense decoder whatever the activation density
    t_fwd_dense, _ = timed(lambda: m(x))
    m.bits_path = "auto"
    del m.decoder.SPARSE_MAX_ACTIVE_FRACTION
    t_fwd_auto, _ = timed(lambda: m(x))
    print(json.dumps(dict(bias_shift_sigmas=shift, active_per_row=round(float(c0.sum()) / B, 1), flagged_rows=flagged,
                          encode_bits_dense_ms=round(t_bits_dense, 3), encode_bits_prefilter_ms=round(t_bits_pref, 3),
                          decode_dense_ms=round(t_dec_dense, 3), decode_sparse_ms=round(t_dec_sparse, 3),
                          forward_dense_ms=round(t_fwd_dense, 3), forward_auto_ms=round(t_fwd_auto, 3),
                          auto_path=m.resolved_bits_path(B))), flush=True)
