#!/usr/bin/env python3
"""A/B in one process (debug library): candidate records of the stationary sweep -- per value (FILT 0, round 1) vs group
records (FILT 1: one maximum and one compare per lane and 16-value tile, hits stored as whole tiles).  Interleaved rounds,
sweep launch timed with HIP events on the launch stream, outputs compared bit for bit.
usage: bench_ab_filter.py [modes ...]   mode = filt + 10 * fill_co   (default: 11 10 1 0)"""
import ctypes as C, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import BinarySAE, _lib, ops
lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
lib.qsae_debug_set_xs_filter.argtypes = [C.c_int]
lib.qsae_debug_set_fill_co.argtypes = [C.c_int]
dev = "cuda:0"
D, H, B = 512, 32768, 65536
g = torch.Generator(device=dev); g.manual_seed(0)
model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(dev).eval()
with torch.no_grad():
    model.decoder.weight.copy_(torch.where(torch.rand_like(model.decoder.weight) > 0.5, 30.0, -30.0))
    model.encoder[0].bias.copy_(torch.randn((H,), device=dev, generator=g) * 0.02)
x = torch.randn((B, D), device=dev, generator=g)
acc = torch.zeros((), dtype=torch.float64, device=dev)
res, ref = {}, None
MODES = [int(a) for a in sys.argv[1:]] or [11, 10, 1, 0]
with torch.no_grad():
    for rnd in range(4):
        for mode in MODES:
            lib.qsae_debug_set_xs_filter(2 if mode % 10 else 0)
            lib.qsae_debug_set_fill_co(mode // 10)
            for _ in range(2):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            ops.kernel_timer.reset()
            ops.sweep_timing(True)
            t0 = time.perf_counter()
            for _ in range(10):
                lat, rec, _ = model(x); ops.sq_err_sum(rec, x, acc)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            ops.sweep_timing(False)
            ms, n, frac = ops.sweep_timing_collect(H)
            res.setdefault(mode, []).append((dt, ms, model.last_flagged_rows))
            if rnd == 0:
                if ref is None:
                    ref = (lat.clone(), rec.clone())
                else:
                    same = bool(torch.equal(ref[0], lat)) and bool(torch.equal(ref[1], rec))
                    print(json.dumps(dict(mode=mode, identical_to_first_mode=same)), flush=True)
            del lat, rec
lib.qsae_debug_set_xs_filter(1)
lib.qsae_debug_set_fill_co(1)
for mode, v in res.items():
    print(json.dumps(dict(filt=mode % 10, fill_co=mode // 10, ms_per_step=[round(t[0], 3) for t in v],
                          sweep_ms=[round(t[1], 3) for t in v], flagged=[t[2] for t in v],
                          median=round(sorted(t[0] for t in v)[len(v) // 2], 3))), flush=True)
