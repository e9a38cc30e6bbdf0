#!/usr/bin/env python3
"""Refinement launch alone (lists from one complete call), timing ablations of its chain loop (debug library)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
lib.qsae_debug_set_refine_ablate.argtypes = [C.c_int]
B = 65536
model = bench.build_model(dev)
x = torch.randn(B, bench.D, device=dev)
lin, pw, dec = model.encoder.linear, model._prefilter_weights(), model.decoder
packed = dec.packed()["packed"]


def submit():
    return ops.binary_forward_prefilter_submit(x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k, packed,
                                               dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=True, slot=0)


def timed(fn, reps=8):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        keep = fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        del keep
    ts.sort()
    return ts[len(ts) // 2]


lib.qsae_debug_set_phases(3, 0)
h = submit()
torch.cuda.synchronize()
del h
lib.qsae_debug_set_phases(2, 0)
names = {0: "complete", 1: "gathers from 8 fixed rows (L1 hits)", 3: "no scalar loads of x", 4: "no LDS transpose",
         5: "no gathers in the main loop", 6: "no scalar loads, no LDS transpose",
         7: "every XCD gathers from 1024 rows of its own (2 MiB: L2 hits)"}
# (ablation 2, "no chains", leaves the keys uninitialised: the ranked indices are garbage and the dense / decode writes
# that follow fault -- never run it with outputs attached)
for abl in (0, 1, 3, 4, 5, 6, 7, 0):
    lib.qsae_debug_set_refine_ablate(abl)
    print(f"refinement + decode, ablation {abl} ({names[abl]}): {timed(submit):.3f} ms", flush=True)
lib.qsae_debug_set_refine_ablate(0)
lib.qsae_debug_set_phases(3, 0)
