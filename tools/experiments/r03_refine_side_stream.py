#!/usr/bin/env python3
"""Item 4(a) of the round-2 verdict as stated: the refinement (+ row decode) of batch i on a SECOND stream, no CU masks, while
the candidate sweep (+ co-resident zero-fill) of batch i+1 runs on the first.  Uses the debug library's phase switch
(qsae_debug_set_phases: 1 = preparation + sweep only, 2 = refinement only) through qsae_prefilter_submit, which never waits
for the GPU; the exact fallback of the 0-1 flagged rows per batch is not run (timing only).  Prints ms per batch for
(A) both phases back to back on one stream, (B) refinement on a side stream, issued after the next sweep, (C) the same with the
refinement issued BEFORE the next sweep."""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from quantizedsae_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
g = torch.Generator(device=dev); g.manual_seed(3)
xs = [torch.randn((65536, 512), device=dev, generator=g) for _ in range(3)]
lin, dec = model.encoder.linear, model.decoder
NS = 3

with _lib.use_library("debug") as lib:
    lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
    pw = dict(zip(("Wq", "meta"), ops.prefilter_pack_w(lin.weight.detach(), lin.bias.detach())))
    packed = dec.packed()["packed"]
    slots = []
    for s in range(NS):
        cargs, keep, outs = ops._decode_prefilter_args(xs[s], lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"],
                                                       model.top_k, ("packed", packed, dec.n_bits, dec.quantization_step),
                                                       dec.bias.detach(), True, 100 + s, "pending")
        word = torch.zeros((1,), dtype=torch.int32).pin_memory()
        slots.append((cargs, keep, outs, word))
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev)

    def phase(mask, s, stream):
        lib.qsae_debug_set_phases(mask, 0)
        with torch.cuda.stream(stream):
            cargs, _k, _o, word = slots[s]
            ops.check(lib.qsae_prefilter_submit(*cargs, C.c_void_p(word.data_ptr()), C.c_void_p(stream.cuda_stream)))

    def run(mode, n=30):
        swept = [torch.cuda.Event() for _ in range(NS)]
        refined = [torch.cuda.Event() for _ in range(NS)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n + 1):
            s, p = i % NS, (i - 1) % NS
            if mode == "A":
                if i < n:
                    phase(1, s, main); phase(2, s, main)
                continue
            if mode == "C" and i > 0:                       # refinement of the previous batch first
                side.wait_event(swept[p]); phase(2, p, side); refined[p].record(side)
            if i < n:
                if i >= NS:
                    main.wait_event(refined[s])              # the slot's lists are free again
                phase(1, s, main); swept[s].record(main)
            if mode == "B" and i > 0:
                side.wait_event(swept[p]); phase(2, p, side); refined[p].record(side)
        torch.cuda.synchronize()
        lib.qsae_debug_set_phases(3, 0)
        return (time.perf_counter() - t0) / n * 1e3

    for mode in ("A", "B", "C", "A", "B", "C"):
        run(mode, 4)
        print(f"mode {mode}: {run(mode):.3f} ms per batch")
