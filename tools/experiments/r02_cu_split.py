#!/usr/bin/env python3
"""Experiment: candidate sweep and refinement of consecutive batches side by side on disjoint CU sets.

The sweep is bound by instruction issue inside a CU; the refinement by the chip's gather bandwidth (it leaves most of
every CU idle).  Two streams created with hipExtStreamCreateWithCUMask -- S CUs for the sweep (+ zero-fill), 256 - S for
the refinement of the previous batch -- would let the two overlap.  This script measures, on the debug library
(qsae_debug_set_phases splits a submit into its sweep and refinement halves):
  * which CUs a masked stream really gets (HW_ID / XCC_ID probe),
  * the refinement alone on R CUs, the sweep alone on S CUs (hidden range in `parts`), and both at once.
"""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402  (build_model: the headline model)

dev = torch.device("cuda:0")
lib = _lib.use_library("debug").__enter__()
lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
lib.qsae_debug_cu_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def probe(stream, label):
    out = torch.zeros(2 * 2048, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        lib.qsae_debug_cu_probe(C.c_void_p(out.data_ptr()), 2048, C.c_void_p(stream.cuda_stream))
    stream.synchronize()
    o = out.cpu().view(-1, 2)
    xcc = o[:, 0] & 0xF
    hw = o[:, 1]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = {}
    for x, *_ in ids:
        per_xcc[x] = per_xcc.get(x, 0) + 1
    print(f"probe {label}: {len(ids)} distinct CUs; per XCD {dict(sorted(per_xcc.items()))}", flush=True)
    return len(ids)


B, D, H, NB = 65536, 512, 32768, 4
torch.manual_seed(0)
model = bench.build_model(dev)
x0 = torch.randn(B, D, device=dev)
x1 = torch.randn(B, D, device=dev)
lin = model.encoder.linear
pw = model._prefilter_weights()
dec = model.decoder
packed = dec.packed()["packed"]
k = model.top_k


def submit(x, slot):
    return ops.binary_forward_prefilter_submit(x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], k, packed,
                                               dec.n_bits, dec.quantization_step, dec.bias.detach(), want_dense=True, slot=slot)


def timed(stream, fn, reps=6):
    ts = []
    for _ in range(reps):
        with torch.cuda.stream(stream):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            keep = fn()
            b.record()
        stream.synchronize()
        ts.append(a.elapsed_time(b))
        del keep
    ts.sort()
    return ts[len(ts) // 2]


main = torch.cuda.current_stream()
probe(main, "default stream")
lib.qsae_debug_set_phases(3, 0)
print(f"baseline, default stream, full submit: {timed(main, lambda: submit(x0, 0)):.3f} ms", flush=True)
lib.qsae_debug_set_phases(1, 0)
print(f"baseline, sweep half only:            {timed(main, lambda: submit(x0, 0)):.3f} ms", flush=True)
lib.qsae_debug_set_phases(2, 0)
print(f"baseline, refinement half only:       {timed(main, lambda: submit(x0, 0)):.3f} ms", flush=True)

for R, parts in ((64, 3), (32, 7), (96, 5), (64, 1)):
    S = 256 - R
    s_sweep = masked_stream(range(0, S))
    s_ref = masked_stream(range(S, 256))
    ns, nr = probe(s_sweep, f"sweep mask {S}"), probe(s_ref, f"refine mask {R}")
    # lists for the refinement-only runs: one complete submit on the refinement stream (same workspace key)
    lib.qsae_debug_set_phases(3, parts)
    with torch.cuda.stream(s_ref):
        h = submit(x0, 1)
    s_ref.synchronize()
    del h
    lib.qsae_debug_set_phases(2, parts)
    t_ref = timed(s_ref, lambda: submit(x0, 1))
    lib.qsae_debug_set_phases(1, parts)
    t_sw = timed(s_sweep, lambda: submit(x1, 2))
    # both at once: wall clock from the host around the two enqueues and the two synchronisations
    walls = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.qsae_debug_set_phases(1, parts)
        with torch.cuda.stream(s_sweep):
            h1 = submit(x1, 2)
        lib.qsae_debug_set_phases(2, parts)
        with torch.cuda.stream(s_ref):
            h2 = submit(x0, 1)
        s_sweep.synchronize()
        s_ref.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        del h1, h2
    walls.sort()
    print(f"S={S} ({ns} CUs) parts={parts}: sweep alone {t_sw:.3f} ms | R={R} ({nr} CUs): refinement alone {t_ref:.3f} ms | "
          f"both at once {walls[len(walls) // 2]:.3f} ms (min {walls[0]:.3f})", flush=True)
    lib.qsae_debug_set_phases(3, 0)
    ops.release_workspaces()
lib.qsae_debug_set_phases(3, 0)
