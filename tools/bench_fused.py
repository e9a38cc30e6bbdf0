#!/usr/bin/env python3
"""Fused encoder+top-k timing under tuning knobs (stagger), interleaved rounds in one process."""
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402

lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_stagger.argtypes = [C.c_int]
B, D, H, k = 65536, 512, 32768, 65
x = torch.randn(B, D, device="cuda:0")
W = (torch.rand(H, D, device="cuda:0") * 2 - 1) * (6.0 / (D + H)) ** 0.5
bias = torch.zeros(H, device="cuda:0")
res = {}
Wp = ops.kperm_rows(W)
xp = ops.kperm_rows(x)
lib.qsae_debug_set_stagger.argtypes = [C.c_int]
lib.qsae_debug_set_sweep_kernel.argtypes = [C.c_int]
for rnd in range(4):
    for kperm in ("dma", "regstage"):
        lib.qsae_debug_set_sweep_kernel(0 if kperm == "dma" else 1)
        a_x, a_W = (xp, Wp)
        ops.encode_topk(a_x, a_W, bias, k, kperm=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.sweep_timing(True)
        e0.record()
        for _ in range(3):
            ops.encode_topk(a_x, a_W, bias, k, kperm=True)
        e1.record(); e1.synchronize()
        ops.sweep_timing(False)
        ms, n, frac = ops.sweep_timing_collect(H)
        res.setdefault(kperm, []).append((ms, e0.elapsed_time(e1) / 3))
lib.qsae_debug_set_sweep_kernel(0)
for kperm, ts in sorted(res.items(), key=lambda kv: str(kv[0])):
    ms = sorted(t[0] for t in ts)[len(ts) // 2]
    tot = sorted(t[1] for t in ts)[len(ts) // 2]
    print(json.dumps(dict(kperm=kperm, sweep_ms=round(ms, 3), sweep_tflops=round(frac * 2.0 * B * D * H / ms / 1e9, 1),
                          encode_topk_total_ms=round(tot, 3))), flush=True)
