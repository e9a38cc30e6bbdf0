"""Config 5 of BASELINE.json: an activation stream in the reference's chunk format through an SAE, row-sharded over ranks.

The reference's consumer of such a stream is ``analyze_dataset`` (scripts/analysis/dynamic_analyze.py:9-24 builds a loader of
32768-row batches over ``HiddenStatesTorchDataset`` chunks of ``[ctx, tok, 512]``, data/dataset.py:16-33;
scripts/analysis/dynamic_analysis.py:345-362 runs the model on every batch and accumulates ``((recon - x)^2).sum()``).  Here
every rank walks its contiguous row range of every chunk (``sharding.shard_rows``: no data-path collective) in device
batches: the host -> device copy of batch i+1 runs on a copy stream while batch i is in the kernels
(``HiddenStatesTorchDataset.iter_batches``), two batches are in flight through ``forward_submit`` / ``result`` where the
model has them (the 4-byte flagged-row count of a batch is read while the next batch's kernels run), and the squared error
is accumulated on the device in fp64 (``qsae_sq_err_sum``) and reduced across ranks once at the end."""
from __future__ import annotations

import time
from typing import Iterable, Optional

import torch

from . import torch_ops as ops
from .sharding import max_over_ranks, reduce_mse


def stream_reconstruction_error(model, datasets: Iterable, device, *, batch_rows: int = 65536, world_size: int = 1,
                                rank: int = 0, in_flight: int = 2, group=None) -> dict:
    """Run ``model`` (any of the five module classes) over this rank's rows of every dataset (``HiddenStatesTorchDataset``-
    like: ``iter_batches``) and return the GLOBAL recon-MSE (all ranks' rows) plus this rank's counters.  ``in_flight`` = 2
    uses the two-call forward where the model has one (BinarySAE / Baseline with compact outputs -- the dense latent is not
    needed for the metric --, QuantizedMatryoshkaSAE), 1 the blocking forward."""
    dev = torch.device(device)
    sq = torch.zeros((), dtype=torch.float64, device=dev)
    rows = batches = 0
    flagged = []
    compact = hasattr(model, "forward_compact")              # BinarySAE / Baseline: (idx, val, reconstruction), no dense latent
    submit = getattr(model, "forward_submit", None) if in_flight > 1 else None

    def reconstruction(outs):
        """The reconstruction in a forward()'s outputs, as the reference's adapters pick it (inference/framework.py:76-111):
        last element of a tuple; the last level where that is a list of levels."""
        rec = outs[-1]
        if compact and len(outs) == 3 and not isinstance(rec, (list, tuple)) and rec.dim() == 0:
            rec = outs[1]                                    # BinarySAE.forward: (latent, reconstruction, polarize_loss)
        return rec[-1] if isinstance(rec, (list, tuple)) else rec

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        pending = None                                       # (handle, batch) of the batch whose result is still out

        def finish(p):
            handle, xb = p
            ops.sq_err_sum(reconstruction(handle.result()), xb, sq)
            flagged.append(int(getattr(model, "last_flagged_rows", 0) or 0))

        for ds in datasets:
            for xb in ds.iter_batches(batch_rows, dev, world_size=world_size, rank=rank):
                if submit is not None:
                    h = submit(xb, slot=batches % 2, want_dense=False) if compact else submit(xb, slot=batches % 2)
                    if pending is not None:
                        finish(pending)
                    pending = (h, xb)
                else:
                    outs = model.forward_compact(xb) if compact else model(xb)
                    ops.sq_err_sum(reconstruction(outs), xb, sq)
                    flagged.append(int(getattr(model, "last_flagged_rows", 0) or 0))
                rows += xb.shape[0]
                batches += 1
        if pending is not None:
            finish(pending)
    torch.cuda.synchronize(dev)
    seconds = max_over_ranks(time.perf_counter() - t0, device=dev, group=group)
    feat = model.input_dim if hasattr(model, "input_dim") else model.encoder.linear.weight.shape[1]
    if hasattr(model, "saes"):
        feat = model.saes[0].input_dim
    local_sq = float(sq.item())
    mse = reduce_mse(sq, rows * feat, group=group) if rows or world_size > 1 else float("nan")
    return {"recon_mse": mse, "rows": rows, "batches": batches, "seconds": seconds, "local_sq_err": local_sq,
            "flagged_rows_per_batch": flagged, "rank": rank, "world_size": world_size}
