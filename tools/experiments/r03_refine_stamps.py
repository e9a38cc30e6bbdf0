#!/usr/bin/env python3
"""Phase stamps of the refinement launch in its headline form (k = 64, row decode attached), complete and with the chain
gathers ablated (debug library).  s_memtime ticks summed over waves, per row."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
lib.qsae_debug_set_refine_ablate.argtypes = [C.c_int]
lib.qsae_debug_set_refine_stamps.argtypes = [C.c_void_p]
B = 65536
model = bench.build_model(dev)
x = torch.randn(B, bench.D, device=dev)
lin, pw, dec = model.encoder.linear, model._prefilter_weights(), model.decoder
packed = dec.packed()["packed"]


def submit():
    return ops.binary_forward_prefilter_submit(x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k, packed,
                                               dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=True, slot=0)


lib.qsae_debug_set_phases(3, 0)
h = submit()
torch.cuda.synchronize()
del h
lib.qsae_debug_set_phases(2, 0)
names = ["list load + keys", "bisection", "cut + survivors", "-", "exact chains", "rank + output", "row decode"]
for abl in (0, 5, 1):
    lib.qsae_debug_set_refine_ablate(abl)
    h = submit(); torch.cuda.synchronize(); del h
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); h = submit(); b.record(); torch.cuda.synchronize(); del h
    plain = a.elapsed_time(b)
    stamps = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.qsae_debug_set_refine_stamps(C.c_void_p(stamps.data_ptr()))
    a.record(); h = submit(); b.record(); torch.cuda.synchronize(); del h
    lib.qsae_debug_set_refine_stamps(None)
    s = stamps.cpu().double() / (B / 64)       # one workgroup in 64 stamps
    print(f"ablation {abl}: {plain:.3f} ms plain, {a.elapsed_time(b):.3f} ms stamped; ticks per row:")
    for nm, v in zip(names, s.tolist()):
        if nm != "-":
            print(f"   {nm:18s} {v:10.0f}  ({100 * v / s.sum().item():4.1f} %)")
    print(f"   {'total':18s} {s.sum().item():10.0f}")
lib.qsae_debug_set_refine_ablate(0)
lib.qsae_debug_set_phases(3, 0)
