#!/usr/bin/env python3
"""Pilot width / rank scan of the in-kernel pilot (debug library): flagged rows per 65536-row batch and step time."""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib  # noqa: E402
import bench  # noqa: E402

lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_pilot.argtypes = [C.c_int, C.c_int]
lib.qsae_debug_set_inkernel_pilot.argtypes = [C.c_int, C.c_int]
dev = torch.device("cuda:0")
m = bench.build_model(dev)
xs = []
for seed in range(3):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    xs.append(torch.randn((65536, 512), device=dev, generator=g))
for div, rank in ((16, 0), (16, 13), (32, 8), (32, 9), (32, 10), (32, 11), (32, 12)):
    lib.qsae_debug_set_pilot(div, 20)
    lib.qsae_debug_set_inkernel_pilot(1, rank)
    flagged = []
    for x in xs:
        m(x); torch.cuda.synchronize()
        flagged.append(m.last_flagged_rows)
    m.last_flagged_rows = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m(xs[0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"pilot H/{div}, rank {rank}: flagged rows {flagged}, {ms:.3f} ms per blocking forward", flush=True)
lib.qsae_debug_set_pilot(16, 20)
lib.qsae_debug_set_inkernel_pilot(1, 0)
