#!/usr/bin/env python3
"""Timing experiment (debug library): the rank / decode launch of batch i on a second, lower-priority stream, so that it runs
beside the select and chain launches of batch i + 1 instead of in front of its sweep.  Results are not checked beyond the
reconstruction sums (the flagged-row bookkeeping is skipped).  ms per step over 40 steps, two batches in flight."""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
model = bench.build_model(dev)
lin, pw, dec = model.encoder.linear, model._prefilter_weights(), model.decoder
packed = dec.packed()["packed"]
B = 65536
xs = [torch.randn(B, bench.D, device=dev) for _ in range(2)]
slots = []
for i in range(2):
    cargs, keep, outs = ops._decode_prefilter_args(xs[i], lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k,
                                                   ("packed", packed, dec.n_bits, dec.quantization_step), dec.bias.detach(), True,
                                                   (991, i), "pending")
    word = torch.zeros((1,), dtype=torch.int32).pin_memory()
    slots.append((cargs, keep, outs, word))


def call(slot, stream):
    cargs, keep, outs, word = slots[slot]
    _lib.check(lib.qsae_prefilter_submit(*cargs, C.c_void_p(word.data_ptr()), C.c_void_p(stream.cuda_stream)))


def run(n, split, prio):
    main = torch.cuda.Stream(priority=-1 if prio else 0)
    side = torch.cuda.Stream(priority=0)
    ev_chain = [torch.cuda.Event() for _ in range(2)]
    ev_rank = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        s = i % 2
        if split:
            if i >= 2:
                main.wait_event(ev_rank[s])                 # the slot's lists are free again
            lib.qsae_debug_set_phases(1 | 2 | 4, 0)
            call(s, main)
            ev_chain[s].record(main)
            side.wait_event(ev_chain[s])
            lib.qsae_debug_set_phases(2 | 8, 0)
            call(s, side)
            ev_rank[s].record(side)
        else:
            lib.qsae_debug_set_phases(3, 0)
            call(s, main)
    torch.cuda.synchronize()
    lib.qsae_debug_set_phases(3, 0)
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    run(6, False, False)
sums = lambda: [float(slots[i][2][3].double().sum()) for i in range(2)]
base = sums()
for rep in range(3):
    a = run(40, False, False)
    b = run(40, True, False)
    c = run(40, True, True)
    print(f"one stream {a:.3f} ms   rank launch on a side stream {b:.3f} ms   with the main stream at high priority {c:.3f} ms   "
          f"sums equal {sums() == base}", flush=True)
