#!/usr/bin/env python3
"""Config 5 of BASELINE.json on the hardware that exists: a hidden-state chunk file in the reference's on-disk format
([num_contexts, tokens_per_context, 512] fp16, data/dataset.py:7-33) streamed through BinarySAE(512, 32768, n_bits=4) in
65536-row device batches, row-sharded per rank (quantizedsae_amd/stream.py), host -> device copies inside the timed loop,
two batches in flight, recon-MSE accumulated on the device and reduced over the ranks.

The pythia-70m layer-3 residuals do not exist offline; two synthetic stand-ins:
  gauss  bell-shaped N(0,1)-like rows, encoder biases 0
  heavy  four of the 512 dimensions at 30x scale, per-row scale spread over 10x, encoder biases != 0, 1 % of the
         encoder rows at 5x norm (quantizedsae_amd/synthetic.py)

    python tools/run_chunk_stream.py --make CHUNK.pt --mode heavy --contexts 1050       # write a chunk
    python tools/run_chunk_stream.py --chunk CHUNK.pt --mode heavy                       # one rank
    python -m torch.distributed.run --nproc-per-node 2 ... tools/run_chunk_stream.py --chunk CHUNK.pt --mode heavy
    python tools/run_chunk_stream.py --chunk CHUNK.pt --mode heavy --stats               # candidates / survivors per row

Every rank prints nothing; rank 0 prints ONE JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

D, H, N_BITS, GAMMA, TOKENS = 512, 32768, 4, 4.0, 250
SEED = 100


def make_chunk(path, mode, contexts, seed=SEED):
    from quantizedsae_amd import synthetic as S
    rows = contexts * TOKENS
    x = S.heavy_tailed_activations(seed, rows, D) if mode == "heavy" else S.activations(seed, rows, D)
    torch.save(torch.from_numpy(x).reshape(contexts, TOKENS, D).to(torch.float16), path)


def build_model(mode, device, seed=SEED):
    from quantizedsae_amd import BinarySAE, synthetic as S
    sd = S.heavy_tailed_binary_sae_params(seed, D, H, N_BITS) if mode == "heavy" else S.binary_sae_params(seed, D, H, N_BITS)
    model = BinarySAE(D, H, gamma=GAMMA, n_bits=N_BITS)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(device).eval()


def list_stats(model, xb):
    """Candidates and survivors per row of one batch, from the lists the candidate sweep leaves in the workspace (debug
    library: same kernels, plus the hook that says where the lists are).  survivors = entries at or above the approximate
    k-th largest minus the row's margin -- the hidden units the refinement evaluates exactly (k of them are mandatory)."""
    from quantizedsae_amd import _lib, ops
    with _lib.use_library("debug") as lib:
        lin = model.encoder.linear
        B, k = xb.shape[0], model.top_k
        Wq, meta = ops.prefilter_pack_w(lin.weight.detach(), lin.bias.detach())
        info = {}
        ops.encode_topk_prefilter(xb, lin.weight.detach(), lin.bias.detach(), Wq, meta, k, want_dense=False, info=info)
        torch.cuda.synchronize()
        offs = [C.c_size_t() for _ in range(5)]
        cap, parts = C.c_int(), C.c_int()
        lib.qsae_debug_prefilter_list_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 5 + [C.POINTER(C.c_int)] * 2
        lib.qsae_debug_prefilter_list_offsets(B, D, H, k, *[C.byref(o) for o in offs], C.byref(cap), C.byref(parts))
        if parts.value != 1:
            return {"flagged_rows": info["flagged_rows"], "note": "hidden range split over several list segments: not summarised"}
        ws = ops._workspace(xb.device, 1)
        cand = ws[offs[0].value: offs[0].value + B * cap.value * 8].view(torch.int32).reshape(B, cap.value, 2)
        cnt = ws[offs[1].value: offs[1].value + B * 4].view(torch.int32).long().clamp(max=cap.value)
        margin = ws[offs[4].value: offs[4].value + B * 4].view(torch.float32)
        vals = cand[:, :, 0].contiguous().view(torch.float32)
        live = torch.arange(cap.value, device=xb.device)[None, :] < cnt[:, None]
        vals = torch.where(live, vals, torch.full_like(vals, float("-inf")))
        kth = torch.topk(vals, k, dim=1).values[:, -1]
        surv = (vals >= (kth - margin)[:, None]).sum(1).float()
        ok = cnt >= k
        return {"flagged_rows": info["flagged_rows"], "candidates_per_row_mean": float(cnt.float().mean()),
                "candidates_per_row_max": int(cnt.max()), "survivors_per_row_mean": float(surv[ok].mean()),
                "survivors_per_row_p99": float(surv[ok].quantile(0.99)), "survivors_per_row_max": float(surv[ok].max()),
                "rows_with_more_than_256_survivors": int((surv > 256).sum()), "rows": B}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make", help="write a synthetic chunk file here and exit")
    ap.add_argument("--chunk", nargs="+", help="chunk file(s) to stream")
    ap.add_argument("--mode", default="gauss", choices=["gauss", "heavy"])
    ap.add_argument("--contexts", type=int, default=1050)
    ap.add_argument("--batch-rows", type=int, default=65536)
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2])
    ap.add_argument("--passes", type=int, default=2, help="passes over the stream; the last one is reported (first = warm-up)")
    ap.add_argument("--stats", action="store_true", help="candidates / survivors per row of the first batch (debug library)")
    args = ap.parse_args()
    if args.make:
        make_chunk(args.make, args.mode, args.contexts)
        return
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("QSAE_BENCH_BACKEND", "nccl")     # gloo: ranks share one card (rehearsal on a one-GPU box)
    device = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    from quantizedsae_amd import data
    from quantizedsae_amd.stream import stream_reconstruction_error
    model = build_model(args.mode, device)
    datasets = [data.HiddenStatesTorchDataset(p) for p in args.chunk]
    out = None
    for _ in range(max(1, args.passes)):
        out = stream_reconstruction_error(model, datasets, device, batch_rows=args.batch_rows, world_size=world, rank=rank,
                                          in_flight=args.in_flight)
    total_rows = sum(len(ds) for ds in datasets)
    doc = {"config": "BinarySAE(512, 32768, gamma=4, n_bits=4), chunk stream", "mode": args.mode, "chunks": len(datasets),
           "chunk_shape": list(datasets[0].data.shape), "chunk_dtype": str(datasets[0].data.dtype), "rows_total": total_rows,
           "world_size": world, "batch_rows": args.batch_rows, "in_flight": args.in_flight,
           "seconds": out["seconds"], "rows_per_s": total_rows / out["seconds"], "recon_mse": out["recon_mse"],
           "rank0_rows": out["rows"], "rank0_batches": out["batches"], "rank0_flagged_rows_per_batch": out["flagged_rows_per_batch"],
           "note": "compact outputs (idx, val, reconstruction); fp16 chunk rows copied host -> device on a copy stream and widened "
                   "there, copies inside the timed loop; time = max over ranks"}
    if args.stats and rank == 0:
        xb = next(iter(datasets[0].iter_batches(args.batch_rows, device, world_size=world, rank=rank)))
        doc["first_batch_lists"] = list_stats(model, xb)
    if rank == 0:
        print(json.dumps(doc), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
