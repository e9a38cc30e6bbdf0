#!/usr/bin/env python3
"""fp16-prefilter encoder+top-k: sweep / total time per sweep kernel variant.

Variants are (tile, rot): tile 2 = activation-stationary sweep (rot = DMA rotation multiplier), 0 = 256 x 256
LDS-tiled GEMM, 1 = 256 x 128; tile >= 10 are timing ablations of the stationary kernel (wrong results)."""
import ctypes as C
import json
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
lib.qsae_debug_set_xstat_rot.argtypes = [C.c_int]
lib.qsae_debug_set_refine_ablate.argtypes = [C.c_int]
lib.qsae_debug_set_pilot.argtypes = [C.c_int, C.c_int]
import os
lib.qsae_debug_set_inkernel_pilot.argtypes = [C.c_int, C.c_int]
if os.environ.get('QSAE_INKERNEL'):
    lib.qsae_debug_set_inkernel_pilot(*[int(v) for v in os.environ['QSAE_INKERNEL'].split(',')])
if os.environ.get('QSAE_PILOT'):
    lib.qsae_debug_set_pilot(*[int(v) for v in os.environ['QSAE_PILOT'].split(',')])

VARIANTS = [(True, 2, 2), (False, 2, 2), (False, 2, 0), (False, 2, 1), (False, 2, 5), (False, 2, 8), (False, 2, 17),
            (False, 11, 2), (False, 11, 0), (False, 0, 0)]
if len(sys.argv) > 1:
    VARIANTS = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    VARIANTS = [(bool(d), t, r) for d, t, r in VARIANTS]
res = {}
for rnd in range(3):
    for dense, tile, rot in VARIANTS:
        lib.qsae_debug_set_prefilter_tile(tile)
        lib.qsae_debug_set_xstat_rot(rot % 10 + 100 * (rot // 100))   # rot >= 100: separate fill pass, >= 1000: old pilot tile
        lib.qsae_debug_set_refine_ablate((rot % 100) // 10)          # tens digit: refine ablation
        ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=dense)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.sweep_timing(True)
        e0.record()
        info = {}
        for _ in range(3):
            ops.encode_topk_prefilter(x, W, bias, Wq, meta, k, want_dense=dense, info=info)
        e1.record(); e1.synchronize()
        ops.sweep_timing(False)
        ms, n, frac = ops.sweep_timing_collect(H)
        res.setdefault((dense, tile, rot), []).append((ms, e0.elapsed_time(e1) / 3, info.get("flagged_rows", -1)))
lib.qsae_debug_set_prefilter_tile(2)
lib.qsae_debug_set_xstat_rot(2)
lib.qsae_debug_set_refine_ablate(0)
for (dense, tile, rot), ts in res.items():
    ms = sorted(t[0] for t in ts)[len(ts) // 2]
    tot = sorted(t[1] for t in ts)[len(ts) // 2]
    print(json.dumps(dict(dense_output=dense, tile=tile, rot=rot, sweep_ms=round(ms, 3), total_ms=round(tot, 3),
                          flagged_rows=ts[-1][2], sweep_fp16_tflops=round(frac * 2.0 * B * D * H / ms / 1e9, 1))), flush=True)
