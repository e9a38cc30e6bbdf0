#!/usr/bin/env python3
"""A/B in one process: speculative first fallback chunk (0 = wait for the flagged-row count, 32, 128 rows)."""
import ctypes as C, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, _lib, ops
lib = _lib.load()
lib.qsae_debug_set_spec_rows.argtypes = [C.c_int]
dev = "cuda:0"
D, H, B = 512, 32768, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
x = torch.randn((B, D), device=dev, generator=g)
acc = torch.zeros((), dtype=torch.float64, device=dev)
res = {}
with torch.no_grad():
    for rnd in range(4):
        for spec in (0, 32, 128):
            lib.qsae_debug_set_spec_rows(spec)
            for _ in range(2):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            res.setdefault(spec, []).append((time.perf_counter() - t0) / 10 * 1e3)
lib.qsae_debug_set_spec_rows(32)
for spec, v in res.items():
    print(json.dumps(dict(spec_rows=spec, ms_per_step=[round(t, 3) for t in v], median=round(sorted(v)[len(v) // 2], 3))))
