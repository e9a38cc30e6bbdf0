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

lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
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


lib.qsae_debug_set_stagger.argtypes = [C.c_int]
lib.qsae_debug_set_sweep.argtypes = [C.c_int]
res = {}
variants = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (0, 2, 0), (0, 4, 0), (0, 8, 0), (0, 4, 64), (0, 0, 64)]   # (ablate, stagger, sweep)
for rnd in range(3):
    for ab, stg, swp in variants:
        lib.qsae_debug_set_stagger(stg)
        lib.qsae_debug_set_sweep(swp)
        run(2, ab)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            run(2, ab)
        b.record(); b.synchronize()
        res.setdefault((ab, stg, swp), []).append(a.elapsed_time(b) / 3)
lib.qsae_debug_set_stagger(0)
lib.qsae_debug_set_sweep(0)
for (ab, stg, swp), ts in sorted(res.items()):
    ms = sorted(ts)[len(ts) // 2]
    print(json.dumps(dict(ablate=ab, stagger=stg, sweep=swp, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))), flush=True)
