#!/usr/bin/env python3
"""Per-phase cycle totals of the activation-stationary sweep (s_memtime stamps, debug build path ABL=5)."""
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
lib.qsae_debug_set_prefilter_tile.argtypes = [C.c_int]
lib.qsae_debug_set_xstat_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros((B // 256, 8, 8), dtype=torch.int64, device="cuda:0")
lib.qsae_debug_set_xstat_stamps(C.c_void_p(stamps.data_ptr()))
names = ["flush", "fill setup", "pass 0 (tile 0 MFMAs + DMA issue + filter of previous tile 1)", "pass 1 (tile 1 MFMAs + fill + filter of tile 0)", "vmcnt wait", "barrier"]
nst = H // 64          # iterations of the stamped sweep loop (the in-kernel pilot's 32 iterations are not stamped)
for label, tile in (("full kernel", 15), ("no hits", 17), ("no filter", 18), ("product form: no fill code, staggered flush", 20)):
    lib.qsae_debug_set_prefilter_tile(tile)
    for _ in range(3):
        try:
            ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=False)
        except Exception as e:      # ablations produce wrong lists; only the stamps matter
            print("ignored:", type(e).__name__)
    torch.cuda.synchronize()
    s = stamps.cpu().double()
    stamps.zero_()
    print("==", label)
    for grp, sl in (("early waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
        tot = 0.0
        row = []
        for i, nm in enumerate(names):
            v = s[:, sl, i].mean().item() / nst
            tot += v
            row.append(f"{nm} {v:.0f}")
        print(f"  {grp}: " + " | ".join(row) + f" | total {tot:.0f}")
lib.qsae_debug_set_prefilter_tile(2)
lib.qsae_debug_set_xstat_stamps(None)
