#!/usr/bin/env python3
"""Refinement as one launch against select / slice-major chains / rank, by batch size (debug library: the form is a switch
there).  Blocking BinarySAE.forward() through the candidate sweep, ms per call (median of 12)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_refine_sliced.argtypes = [C.c_int]
kind = sys.argv[1] if len(sys.argv) > 1 else "binary"          # binary (headline) | soft (unpolarised decoder: fp32 table) | baseline (top-32, fp32 table)
if kind == "binary":
    model = bench.build_model(dev)
    model.latent_path = "prefilter"
    model.decoder.packed()
elif kind == "soft":
    import warnings
    from quantizedsae_amd import BinarySAE
    warnings.simplefilter("ignore")
    model = BinarySAE(bench.D, bench.H, gamma=bench.GAMMA, n_bits=bench.N_BITS).to(dev).eval()
    with torch.no_grad():
        model.decoder.weight.normal_(0, 2.0)
else:
    from quantizedsae_amd import BaselineSparseAutoencoder
    model = BaselineSparseAutoencoder(bench.D, bench.H).to(dev).eval()
if len(sys.argv) > 2:                                              # another k (the classes derive theirs from the width / the constructor)
    k = int(sys.argv[2])
    if kind == "baseline":
        model.k = k
    else:
        type(model).top_k = property(lambda self: k)
print(kind, "k =", getattr(model, "top_k", getattr(model, "k", None)))
for B in ((2048, 4096, 8192, 16384, 32768, 65536, 131072) if kind == "binary" and len(sys.argv) < 3 else (8192, 16384, 65536)):
    x = torch.randn((B, 512), device=dev)
    res = {}
    for form in (0, 2, 0, 2):
        lib.qsae_debug_set_refine_sliced(form)
        for _ in range(3):
            out = model(x)
        ts = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = model(x); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        res.setdefault(form, []).append(ts[len(ts) // 2])
        del out
    print(f"B = {B:6d}: one launch {min(res[0]):7.3f} ms   three launches {min(res[2]):7.3f} ms", flush=True)
lib.qsae_debug_set_refine_sliced(1)
