#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels on the full BinarySAE shape (GPU box only).
Interleaved rounds in one process (cdna guide rule 24); prints one JSON line per measurement."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from quantizedsae_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--H", type=int, default=32768)
    ap.add_argument("--D", type=int, default=512)
    ap.add_argument("--k", type=int, default=65)
    ap.add_argument("--configs", default="1,2,3")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--sweeps", default="1,2,4,8,16,64")
    ap.add_argument("--only-gemm", action="store_true")
    a = ap.parse_args()
    B, H, D, k = a.B, a.H, a.D, a.k
    dev = "cuda:0"
    lib = _lib.use_library("debug").__enter__()   # tools run against libqsae_hip_debug.so (qsae_debug_* switches)
    lib.qsae_debug_set_gemm_config.argtypes = [C.c_int]
    lib.qsae_debug_set_sweep.argtypes = [C.c_int]
    lib.qsae_debug_set_topk_path.argtypes = [C.c_int]
    torch.manual_seed(0)
    x = torch.randn(B, D, device=dev)
    bound = (6.0 / (D + H)) ** 0.5
    W = (torch.rand(H, D, device=dev) * 2 - 1) * bound
    bias = torch.zeros(H, device=dev)
    out = torch.empty(B, H, device=dev)
    flops = 2.0 * B * D * H
    res = []
    for rnd in range(2):
        for cfg in [int(c) for c in a.configs.split(",")]:
            lib.qsae_debug_set_gemm_config(cfg)
            med, best = timeit(lambda: ops.encode_dense(x, W, bias, ops.ACT_NONE, out=out), a.iters)
            res.append(dict(kernel="encode_dense", cfg=cfg, round=rnd, ms_med=med, ms_min=best,
                            tflops_med=flops / med / 1e9, tflops_best=flops / best / 1e9))
            print(json.dumps(res[-1]), flush=True)
    lib.qsae_debug_set_gemm_config(0)
    for sweep in [int(v) for v in a.sweeps.split(",") if v]:
        lib.qsae_debug_set_sweep(sweep)
        med, best = timeit(lambda: ops.encode_dense(x, W, bias, ops.ACT_NONE, out=out), a.iters)
        print(json.dumps(dict(kernel="encode_dense", cfg="auto", sweep=sweep, ms_med=med, ms_min=best,
                              tflops_med=flops / med / 1e9)), flush=True)
    lib.qsae_debug_set_sweep(0)
    for path in (2, 1):
        lib.qsae_debug_set_topk_path(path)
        med, best = timeit(lambda: ops.encode_topk(x, W, bias, k), 3, 1)
        print(json.dumps(dict(kernel="encode_topk", path={1: "chunked", 2: "fused"}[path], ms_med=med, ms_min=best,
                              tflops=flops / med / 1e9)), flush=True)
    lib.qsae_debug_set_topk_path(0)
    if a.only_gemm:
        return
    # top-k in place (needs fresh latents each time: time includes only the kernel, data stays the same shape)
    ops.encode_dense(x, W, bias, ops.ACT_NONE, out=out)
    lat = out.clone() if B * H * 4 < 60e9 else out
    med, best = timeit(lambda: ops.topk_rows(out, k, zero_rest=False), a.iters)
    print(json.dumps(dict(kernel="topk_rows(no rewrite)", ms_med=med, ms_min=best, GBps=B * H * 4 / med / 1e6)), flush=True)

    def topk_rw():
        out.copy_(lat)
        ops.topk_rows(out, k, zero_rest=True)
    medc, _ = timeit(lambda: out.copy_(lat), a.iters)
    med, best = timeit(topk_rw, a.iters)
    print(json.dumps(dict(kernel="topk_rows(rewrite) minus copy", ms_med=med - medc, copy_ms=medc,
                          GBps=2 * B * H * 4 / max(med - medc, 1e-3) / 1e6)), flush=True)
    idx, val = ops.topk_rows(out, k, zero_rest=True)
    logits = (torch.randint(0, 2, (H, D * 4), device=dev).float() * 2 - 1) * 30.0
    packed, pol = ops.pack_binary(logits, D, 4)
    med, best = timeit(lambda: ops.pack_binary(logits, D, 4), 3)
    print(json.dumps(dict(kernel="pack_binary", ms_med=med)), flush=True)
    dbias = torch.zeros(D, device=dev)
    med, best = timeit(lambda: ops.decode_binary_sparse(idx, val, packed, D, 4, 0.5, dbias), a.iters)
    print(json.dumps(dict(kernel="decode_binary_sparse", ms_med=med, ms_min=best,
                          gatherGBps=B * k * 256 / med / 1e6)), flush=True)
    med, best = timeit(lambda: ops.densify(idx, val, H, out=out), a.iters)
    print(json.dumps(dict(kernel="densify(memset+scatter)", ms_med=med, GBps=B * H * 4 / med / 1e6)), flush=True)
    recon = ops.decode_binary_sparse(idx, val, packed, D, 4, 0.5, dbias)
    med, best = timeit(lambda: ops.sq_err_sum(recon, x), a.iters)
    print(json.dumps(dict(kernel="sq_err_sum", ms_med=med)), flush=True)


if __name__ == "__main__":
    main()
