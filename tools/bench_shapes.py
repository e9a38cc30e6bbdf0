#!/usr/bin/env python3
"""encode_dense TFLOP/s over problem shapes (cache-resident vs streaming) -- diagnosis helper."""
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402

lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_gemm_config.argtypes = [C.c_int]
dev = "cuda:0"
D = 512


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


shapes = [(4096, 4096), (8192, 8192), (16384, 8192), (16384, 16384), (32768, 16384), (65536, 32768), (65536, 2048),
          (2048, 32768)]
for cfg in (2, 1):
    lib.qsae_debug_set_gemm_config(cfg)
    for B, H in shapes:
        x = torch.randn(B, D, device=dev)
        W = (torch.rand(H, D, device=dev) * 2 - 1) * 0.0134
        out = torch.empty(B, H, device=dev)
        reps = max(1, int(2e12 / (2.0 * B * H * D)))
        reps = min(reps, 50)

        def run():
            for _ in range(reps):
                ops.encode_dense(x, W, None, ops.ACT_NONE, out=out)
        ms = timeit(run) / reps
        print(json.dumps(dict(cfg=cfg, B=B, H=H, ms=round(ms, 4), tflops=round(2.0 * B * H * D / ms / 1e9, 1),
                              out_GB=B * H * 4 / 1e9)), flush=True)
        del x, W, out
