#!/usr/bin/env python3
"""Run the encoder GEMM with ablate = 0, 1, 2 once each (for rocprofv3 --pmc comparisons)."""
import ctypes as C
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
for ab in (0, 1, 2):
    for _ in range(2):
        assert f(x.data_ptr(), W.data_ptr(), B, D, H, out.data_ptr(), 2, ab, None) == 0
torch.cuda.synchronize()
print("done", flush=True)
