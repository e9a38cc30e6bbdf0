#!/usr/bin/env python3
"""fp16-prefilter encoder+top-k: sweep / total time with and without the fused dense-latent zero-fill."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import ops  # noqa: E402

B, D, H, k = 65536, 512, 32768, 65
x = torch.randn(B, D, device="cuda:0")
W = (torch.rand(H, D, device="cuda:0") * 2 - 1) * (6.0 / (D + H)) ** 0.5
bias = torch.zeros(H, device="cuda:0")
Wq, meta = ops.prefilter_pack_w(W, bias)
res = {}
for rnd in range(3):
    for dense in (True, False):
        ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=dense)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.sweep_timing(True)
        e0.record()
        for _ in range(3):
            ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=dense)
        e1.record(); e1.synchronize()
        ops.sweep_timing(False)
        ms, n, frac = ops.sweep_timing_collect(H)
        res.setdefault(dense, []).append((ms, e0.elapsed_time(e1) / 3))
for dense, ts in res.items():
    ms = sorted(t[0] for t in ts)[len(ts) // 2]
    tot = sorted(t[1] for t in ts)[len(ts) // 2]
    print(json.dumps(dict(dense_output=dense, sweep_ms=round(ms, 3), total_ms=round(tot, 3),
                          sweep_fp16_tflops=round(frac * 2.0 * B * D * H / ms / 1e9, 1))), flush=True)
