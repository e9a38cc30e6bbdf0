"""Reader for ``.safetensors`` checkpoints that needs nothing but the standard library and torch
(reference: src/quantized_sae/data/load_baseline.py:5-53, the fallback of inference/framework.py:236-260 when the
``safetensors`` package is absent).

File layout: 8 bytes little-endian header length N, N bytes of JSON ``{name: {"dtype", "shape", "data_offsets":
[begin, end]}, "__metadata__": {...}}``, then the tensor bytes; offsets are relative to the end of the header.
Unlike the reference's reader, which silently reinterprets unknown dtypes as fp32, this one checks what it reads:
header length against the file size, offsets against the payload, byte counts against shape x dtype, and refuses dtypes
it does not know.  The payload is memory-mapped, every tensor is copied out (the result owns its memory).
"""
from __future__ import annotations

import json
import mmap
import struct
from collections import OrderedDict
from math import prod
from typing import Dict

import torch

_DTYPES = {
    "F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16,
    "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool,
}


def load_safetensors(filepath) -> "OrderedDict[str, torch.Tensor]":
    """name -> CPU tensor for every entry of the file, in header order."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    with open(filepath, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{filepath}: not a safetensors file (shorter than its 8-byte length prefix)")
        (n_header,) = struct.unpack("<Q", head)
        f.seek(0, 2)
        size = f.tell()
        if n_header > size - 8:
            raise ValueError(f"{filepath}: header length {n_header} exceeds the file size {size}")
        f.seek(8)
        try:
            header = json.loads(f.read(n_header).decode("utf-8"))
        except (UnicodeDecodeError, json.JSONDecodeError) as e:
            raise ValueError(f"{filepath}: header is not valid JSON") from e
        if not isinstance(header, dict):
            raise ValueError(f"{filepath}: header is not a JSON object")
        base = 8 + n_header
        payload = size - base
        entries = [(k, v) for k, v in header.items() if k != "__metadata__"]
        if not entries:
            return out
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            for name, info in entries:
                try:
                    dtype = _DTYPES[info["dtype"]]
                    shape = [int(s) for s in info["shape"]]
                    begin, end = (int(v) for v in info["data_offsets"])
                except (KeyError, TypeError, ValueError) as e:
                    raise ValueError(f"{filepath}: malformed or unsupported entry for '{name}': {info!r}") from e
                nbytes = prod(shape) * torch.empty((), dtype=dtype).element_size()
                if not (0 <= begin <= end <= payload) or end - begin != nbytes:
                    raise ValueError(f"{filepath}: '{name}' claims bytes [{begin}, {end}) for shape {shape} {info['dtype']} "
                                     f"({nbytes} bytes) in a payload of {payload}")
                if nbytes == 0:
                    out[name] = torch.empty(shape, dtype=dtype)
                    continue
                buf = bytearray(mm[base + begin: base + end])                # a private, writable copy
                out[name] = torch.frombuffer(buf, dtype=dtype).reshape(shape)
    return out
