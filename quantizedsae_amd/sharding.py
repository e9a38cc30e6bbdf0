"""Row sharding of an activation stream across the GPUs of one node.

Rows are independent, so the path shards with no data-path collective: rank r owns one
contiguous row range of every chunk, weights (~72 MiB packed + fp32) are replicated, every rank
writes its own outputs.  The only cross-rank quantity is the recon-MSE pair
(sum of squared error as fp64, element count), reduced once at the end.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch


def shard_rows(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of rank `rank`; sizes differ by at most one row, earlier ranks
    take the remainder, empty ranges are legal (more ranks than rows)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if n_rows < 0:
        raise ValueError("n_rows must be >= 0")
    base, rem = divmod(n_rows, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def iter_chunk_shards(chunk_rows, world_size: int, rank: int) -> Iterator[Tuple[int, int, int]]:
    """For a stream of chunks with the given row counts (e.g. the reference's hidden-state dumps of
    [ctx, tok, 512] flattened to ctx*tok rows, data/dataset.py:16-33) yield
    (chunk_index, start, stop) of this rank's slice of every chunk."""
    for i, n in enumerate(chunk_rows):
        s, e = shard_rows(int(n), world_size, rank)
        yield i, s, e


def reduce_mse(sq_err_sum: torch.Tensor, n_elements: int, group=None) -> float:
    """Global MSE from per-rank (sum of squared error, element count).  Uses a 2-element fp64
    all-reduce when torch.distributed is initialised (RCCL on GPU tensors, gloo on CPU), otherwise
    the local pair.  The reference's single-process recipe: scripts/analysis/dynamic_analysis.py:86-100."""
    import torch.distributed as dist
    pair = torch.stack([sq_err_sum.detach().to(torch.float64).reshape(()),
                        torch.tensor(float(n_elements), dtype=torch.float64, device=sq_err_sum.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo":
            pair = pair.cpu()
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
    total, count = float(pair[0].item()), float(pair[1].item())
    if count == 0:
        raise ValueError("no elements")
    return total / count


def max_over_ranks(value: float, device: Optional[torch.device] = None, group=None) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    if dist.get_backend(group) == "gloo":
        device = None
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
