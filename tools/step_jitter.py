#!/usr/bin/env python3
"""Per-step wall time of the headline forward over many steps (each step synchronised): looks for outliers, e.g. a
sweep workgroup kept off its CU by early fill waves."""
import ctypes as C, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, _lib, ops
dev = "cuda:0"
D, H, B = 512, 32768, 65536
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
x = torch.randn((B, D), device=dev)
acc = torch.zeros((), dtype=torch.float64, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ts, out = [], []
import os
lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
if os.environ.get('QSAE_FILL_CO') is not None:
    lib.qsae_debug_set_fill_co.argtypes = [C.c_int]
    lib.qsae_debug_set_fill_co(int(os.environ['QSAE_FILL_CO']))
with torch.no_grad():
    for i in range(n + 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        if os.environ.get('QSAE_COMPACT'):
            _i, _v, rec = model.forward_compact(x)
        else:
            lat, rec, _ = model(x)
        ops.sq_err_sum(rec, x, acc)
        e1.record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 5:
            ts.append(dt)
            if dt > 6.0:
                out.append(dict(step=i, wall_ms=round(dt, 3), gpu_ms=round(e0.elapsed_time(e1), 3), flagged=int(model.last_flagged_rows)))
for o in out:
    print(json.dumps(o))
ts.sort()
print(json.dumps(dict(steps=n, min=round(ts[0], 3), p50=round(ts[n // 2], 3), p90=round(ts[int(n * 0.9)], 3),
                      p99=round(ts[int(n * 0.99)], 3), max=round(ts[-1], 3))))
