#!/usr/bin/env python3
"""Run each encoder-GEMM configuration a few times at the full shape (for rocprofv3 --pmc / --kernel-trace)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402

B, D, H = 65536, 512, 32768
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_gemm_config.argtypes = [C.c_int]
x = torch.randn(B, D, device="cuda:0")
W = (torch.rand(H, D, device="cuda:0") * 2 - 1) * (6.0 / (D + H)) ** 0.5
bias = torch.zeros(H, device="cuda:0")
out = torch.empty(B, H, device="cuda:0")
for cfg in cfgs:
    lib.qsae_debug_set_gemm_config(cfg)
    for _ in range(reps):
        ops.encode_dense(x, W, bias, ops.ACT_NONE, out=out)
torch.cuda.synchronize()
print("done")
