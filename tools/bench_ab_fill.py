#!/usr/bin/env python3
"""A/B in one process: zero-fill of the dense latent inside the sweep (0) vs by a co-resident fill kernel (1)."""
import ctypes as C, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, _lib, ops
lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_fill_co.argtypes = [C.c_int]
dev = "cuda:0"
D, H, B = 512, 32768, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
with torch.no_grad():
    model.decoder.weight.copy_(torch.where(torch.rand_like(model.decoder.weight) > 0.5, 30.0, -30.0))
x = torch.randn((B, D), device=dev, generator=g)
acc = torch.zeros((), dtype=torch.float64, device=dev)
res, outs = {}, {}
MODES = [int(a) for a in sys.argv[1:]] or [0, 1]
with torch.no_grad():
    for rnd in range(4):
        for mode in MODES:
            lib.qsae_debug_set_fill_co(mode)
            for _ in range(2):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            ops.sweep_timing(True)
            t0 = time.perf_counter()
            for _ in range(10):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            ops.sweep_timing(False)
            ms, n, frac = ops.sweep_timing_collect(H)
            res.setdefault(mode, []).append((dt, ms))
            if rnd == 0:
                if mode == MODES[0]:
                    outs[mode] = (lat.clone(), rec.clone())
                else:
                    assert torch.equal(outs[MODES[0]][0], lat) and torch.equal(outs[MODES[0]][1], rec)
lib.qsae_debug_set_fill_co(0)
for mode, v in res.items():
    print(json.dumps(dict(fill_co=mode, ms_per_step=[round(t[0], 3) for t in v], sweep_ms=[round(t[1], 3) for t in v],
                          median=round(sorted(t[0] for t in v)[len(v) // 2], 3))))
