"""Hidden-state chunk datasets (reference: src/quantized_sae/data/dataset.py:7-102).

A chunk is a ``.pt`` file holding one tensor ``[num_contexts, tokens_per_context, feature_dim]`` (feature_dim
512 for pythia-70m layer 3); a sample is one token's row, ``.float()``.  The classes keep the reference's
constructor arguments, ``__len__`` / ``__getitem__`` and the quantizer method names.  Two things differ by
design: the file is read with ``weights_only=True`` (a tensor file needs no unpickling of code), and besides
the per-sample interface there is a batch interface for the GPU path -- ``rows()`` / ``iter_batches()`` hand
out contiguous row ranges (row-sharded per rank) and ``quantize_batch`` runs the n-bit quantizer
as one HIP kernel (``qsae_quantize_bits``) instead of per sample on the host.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch
from torch.utils.data import Dataset

from . import torch_ops as ops
from .sharding import shard_rows


def _load_chunk(file_path) -> torch.Tensor:
    data = torch.load(file_path, map_location="cpu", weights_only=True)
    if not isinstance(data, torch.Tensor) or data.dim() != 3:
        raise ValueError(f"{file_path}: expected one tensor [num_contexts, tokens_per_context, feature_dim]")
    return data


class HiddenStatesTorchDataset(Dataset):
    """One chunk file, one 512-d fp32 sample per token (dataset.py:7-33)."""

    def __init__(self, file_path, transform=None):
        self.data = _load_chunk(file_path)
        self.transform = transform
        num_contexts, tokens_per_context, feature_dim = self.data.shape
        self.cum_sizes = num_contexts * tokens_per_context
        self.files_info = (file_path, num_contexts, tokens_per_context, feature_dim)

    def __len__(self):
        return self.cum_sizes

    def __getitem__(self, idx):
        context_idx = idx // self.files_info[2]
        token_idx = idx % self.files_info[2]
        return self.data[context_idx, token_idx, :].float()

    # ---- batch interface ---------------------------------------------------------------------------
    def rows(self, start: int = 0, stop: Optional[int] = None) -> torch.Tensor:
        """fp32 rows [stop - start, feature_dim] of the chunk flattened row-major (host tensor)."""
        flat = self.data.reshape(self.cum_sizes, self.files_info[3])
        return flat[start:stop].float()

    def iter_batches(self, batch_rows: int, device, world_size: int = 1, rank: int = 0,
                     prefetch: bool = True) -> Iterator[torch.Tensor]:
        """This rank's contiguous slice of the chunk (``sharding.shard_rows``) in device batches.

        The stored dtype is shipped (fp16 dumps: half the bytes over PCIe) and widened on the device; fp16/bf16 -> fp32
        is exact, so the values equal the reference's host-side ``.float()``.  With ``prefetch`` the chunk is page-
        locked once and batch i+1 is copied on a second stream while batch i is being processed (the forward pass
        has a host round trip per batch, so a copy issued after it would start when the GPU is already idle)."""
        s, e = shard_rows(self.cum_sizes, world_size, rank)
        dev = torch.device(device)
        if not (prefetch and dev.type == "cuda"):
            flat = self.data.reshape(self.cum_sizes, self.files_info[3])
            for a in range(s, e, batch_rows):
                yield flat[a:min(a + batch_rows, e)].to(dev, non_blocking=True).float()
            return
        if not self.data.is_pinned():
            self.data = self.data.pin_memory()
        flat = self.data.reshape(self.cum_sizes, self.files_info[3])
        copy_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)

        def issue(a):
            with torch.cuda.stream(copy_stream):
                t = flat[a:min(a + batch_rows, e)].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return t, ev

        pending = issue(s) if s < e else None
        for a in range(s, e, batch_rows):
            t, ev = pending
            pending = issue(a + batch_rows) if a + batch_rows < e else None
            cur.wait_event(ev)
            t.record_stream(cur)
            yield t.float()


class HiddenStatesTorchDatasetInBinary(HiddenStatesTorchDataset):
    """Samples as n-bit codes, LSB-first 0/1 floats of length feature_dim * n_bits (dataset.py:35-102)."""

    def __init__(self, file_path, gamma=4, n_bits=4, transform=None):
        super().__init__(file_path, transform)
        self.gamma = gamma
        self.n_bits = n_bits
        self.shift_factor = 2 ** (self.n_bits - 1)
        self.scale_factor = 2 ** (self.n_bits - 1) / (self.gamma + 1e-5)

    def __getoriginalitem__(self, idx):
        return HiddenStatesTorchDataset.__getitem__(self, idx)

    def __getitem__(self, idx):
        return self.quantize_signed(self.__getoriginalitem__(idx))

    def _quantize_host(self, sample: torch.Tensor, signed: bool) -> torch.Tensor:
        # per-sample interface of the reference: a single row through the same kernel
        dev = torch.device("cuda", torch.cuda.current_device())
        out = ops.quantize_bits(sample.reshape(1, -1).to(dev), self.n_bits, self.scale_factor, signed=signed)
        return out.reshape(-1).to(sample.device)

    def quantize(self, sample):
        """Unsigned code of (x * scale * 2 + 2^(n-1)) (dataset.py:76-87)."""
        return self._quantize_host(sample, signed=False)

    def quantize_signed(self, sample):
        """Two's-complement code of x * scale (dataset.py:89-102)."""
        return self._quantize_host(sample, signed=True)

    def quantize_batch(self, x: torch.Tensor, signed: bool = True) -> torch.Tensor:
        """[B, feature_dim] device rows -> [B, feature_dim * n_bits] bit floats, one kernel."""
        return ops.quantize_bits(x, self.n_bits, self.scale_factor, signed=signed)
