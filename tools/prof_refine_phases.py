#!/usr/bin/env python3
"""Per-phase cycle totals of refine_topk_kernel (s_memtime stamps, summed over waves)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402

B, D, H, k = 65536, 512, 32768, 65
x = torch.randn(B, D, device="cuda:0")
W = (torch.rand(H, D, device="cuda:0") * 2 - 1) * (6.0 / (D + H)) ** 0.5
bias = torch.zeros(H, device="cuda:0")
Wq, meta = ops.prefilter_pack_w(W, bias)
lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_refine_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(8, dtype=torch.int64, device="cuda:0")
ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=False)
torch.cuda.synchronize()
lib.qsae_debug_set_refine_stamps(C.c_void_p(stamps.data_ptr()))
ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=False)
torch.cuda.synchronize()
lib.qsae_debug_set_refine_stamps(None)
names = ["list load + keys", "bisection", "cut + survivors", "x row -> LDS", "exact chains", "rank + output"]
s = stamps.cpu().double() / (B / 64)       # one workgroup in 64 stamps
for nm, v in zip(names, s.tolist()):
    print(f"{nm:18s} {v:10.0f} cycles/row")
print(f"{'total':18s} {s.sum().item():10.0f} cycles/row")
