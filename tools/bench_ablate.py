#!/usr/bin/env python3
"""A/B of the encoder GEMM with pipeline stages removed (diagnosis; interleaved rounds, one process)."""
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib  # noqa: E402

lib = _lib.load()
f = lib.qsae_debug_encode_ablate
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
B, D, H = 32768, 512, 16384
x = torch.randn(B, D, device="cuda:0")
W = (torch.rand(H, D, device="cuda:0") * 2 - 1) * 0.0134
out = torch.empty(B, H, device="cuda:0")
flops = 2.0 * B * D * H


def run(cfg, ab):
    rc = f(x.data_ptr(), W.data_ptr(), B, D, H, out.data_ptr(), cfg, ab, None)
    assert rc == 0


res = {}
for rnd in range(3):
    for cfg in (2,):
        for ab in (0, 1, 2):
            run(cfg, ab)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                run(cfg, ab)
            b.record(); b.synchronize()
            res.setdefault((cfg, ab), []).append(a.elapsed_time(b) / 3)
for (cfg, ab), ts in sorted(res.items()):
    ms = sorted(ts)[len(ts) // 2]
    print(json.dumps(dict(cfg=cfg, ablate=ab, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))), flush=True)
