"""Host-side logic and the C-ABI surface; runs without a GPU."""
import ctypes
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle
from quantizedsae_amd import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,
                              SparseAutoencoder, TernarySparseAutoencoder, _lib, sharding)
from quantizedsae_amd.inference import framework as F
from quantizedsae_amd.sae.quantized_matryoshka import nested_sizes

ROOT = Path(__file__).resolve().parents[1]


# ---- C ABI --------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    declared = ge.declared_symbols()
    assert len(declared) >= 20 and "qsae_encode_dense" in declared
    assert _lib.LIB_PATH.exists(), "build with `python -m quantizedsae_amd.build`"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared          # the ctypes table binds exactly the header
    assert _lib.load().qsae_abi_version() == _lib.ABI_VERSION == 4


def test_product_library_has_no_debug_surface():
    """The shipped library exports exactly the C ABI of include/qsae.h: no qsae_debug_* switch, no process-wide
    tunable.  Those live in libqsae_hip_debug.so (same sources, -DQSAE_DEBUG_BUILD), which tools/ and a few tests load
    explicitly; it must offer the whole ABI as well."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    from quantizedsae_amd import build
    declared = set(ge.declared_symbols())
    prod = [s_ for s_ in build.exported_symbols(_lib.LIB_PATH) if s_.startswith("qsae_")]
    assert set(prod) == declared, sorted(set(prod) ^ declared)
    assert not [s_ for s_ in prod if "debug" in s_]
    assert _lib.DEBUG_LIB_PATH.exists(), "build with `python -m quantizedsae_amd.build`"
    dbg = {s_ for s_ in build.exported_symbols(_lib.DEBUG_LIB_PATH) if s_.startswith("qsae_")}
    assert declared <= dbg and any(s_.startswith("qsae_debug_") for s_ in dbg)
    with _lib.use_library("debug") as lib:
        assert lib.qsae_abi_version() == 4 and _lib.load() is lib
    assert _lib.load() is not lib


def test_host_side_entry_points():
    lib = _lib.load()
    assert lib.qsae_binary_row_bytes(512, 4) == 256
    assert lib.qsae_binary_row_bytes(512, 8) == 512
    assert lib.qsae_binary_row_bytes(20, 3) == 12          # 20 nibbles = 80 bits -> 3 dwords
    assert lib.qsae_binary_row_bytes(0, 4) == _lib.ERR_INVALID_ARG
    assert lib.qsae_binary_row_bytes(512, 9) == _lib.ERR_INVALID_ARG
    for H, n in [(32768, 4), (1000, 4), (4096, 1), (1100, 4), (97, 3)]:
        arr = (ctypes.c_int32 * n)()
        assert lib.qsae_matryoshka_sizes(H, n, ctypes.cast(arr, ctypes.c_void_p)) == 0
        assert list(arr) == oracle.matryoshka_sizes(H, n) == nested_sizes(H, n)
    assert lib.qsae_matryoshka_sizes(0, 4, None) == _lib.ERR_INVALID_ARG
    assert b"invalid argument" in lib.qsae_last_error()
    # B == 0 is a no-op that needs no device
    assert lib.qsae_encode_dense(None, None, None, 0, 512, 32768, 0, None, 32768, None) == 0
    assert lib.qsae_topk_rows(None, 32768, 0, 32768, 65, None, None, 1, None) == 0
    # argument validation happens before any HIP call
    assert lib.qsae_encode_dense(None, None, None, 4, 512, 32768, 0, None, 32768, None) == _lib.ERR_INVALID_ARG
    assert lib.qsae_topk_rows(ctypes.c_void_p(16), 32768, 1, 32768, 300, ctypes.c_void_p(16), ctypes.c_void_p(16),
                              1, None) == _lib.ERR_UNSUPPORTED
    assert lib.qsae_encode_topk_workspace_bytes(512, 512, 32768, 65) == 512 * 32768 * 4        # chunked form
    assert lib.qsae_encode_topk_workspace_bytes(1500, 512, 2048, 65) == 1024 * 2048 * 4
    big = lib.qsae_encode_topk_workspace_bytes(65536, 512, 32768, 65)                          # fused form
    assert 65536 * (2048 * 4 + 1024 * 8) < big < 2 * 1024 ** 3
    assert lib.qsae_encode_topk_workspace_bytes(0, 512, 32768, 65) == 0


def test_no_cpu_fallback():
    m = BinarySAE(64, 512, gamma=4.0, n_bits=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 64))
    with pytest.raises(RuntimeError):
        BaselineSparseAutoencoder(64, 512)(torch.zeros(2, 64))
    with pytest.raises(RuntimeError):
        TernarySparseAutoencoder(64, 512)(torch.zeros(2, 64))
    with pytest.raises(RuntimeError):
        QuantizedMatryoshkaSAE(64, 512, 32, n_bits=4)(torch.zeros(2, 64))
    with pytest.raises(TypeError):
        m([1, 2, 3])


# ---- module face ----------------------------------------------------------------------------------
def _shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_contract():
    D, H, n = 512, 2048, 4
    assert _shapes(BinarySAE(D, H, gamma=4.0, n_bits=n)) == {
        "encoder.0.weight": (H, D), "encoder.0.bias": (H,), "decoder.weight": (H, D * n), "decoder.bias": (D,)}
    assert _shapes(BaselineSparseAutoencoder(D, H)) == {
        "encoder.0.weight": (H, D), "encoder.0.bias": (H,), "decoder.weight": (D, H), "decoder.bias": (D,)}
    assert _shapes(TernarySparseAutoencoder(D, H)) == {
        "encoder.0.weight": (H, D), "encoder.0.bias": (H,), "decoder.weight": (D, H), "decoder.mask": (D, H)}
    assert _shapes(QuantizedMatryoshkaSAE(D, H, 32, n_bits=n)) == {
        "encoder.0.weight": (H, D), "encoder.0.bias": (H,), "decoder.weight": (H, D),
        "decoder.weight_mirror": (H, D), "decoder.bias": (D,)}
    rq = ResidualQuantizedSAE(D, H, 32, n_bits=n)
    assert rq.sae_hidden_dims == [256, 256, 512, 1024]
    keys = set(rq.state_dict())
    for i in range(n):
        for k in ("encoder.0.weight", "encoder.0.bias", "decoder.weight", "decoder.weight_mirror", "decoder.bias"):
            assert f"saes.{i}.{k}" in keys


def test_constructor_attributes():
    b = BinarySAE(512, 32768, gamma=4.0, n_bits=4)
    assert (b.n_bits, b.input_dim, b.hidden_dim, b.k, b.top_k) == (4, 512, 32768, 0.002, 65)
    assert b.decoder.quantization_step == 0.5 and b.decoder.in_features == 32768 and b.decoder.n_bits == 4
    assert BinarySAE(512, 32768, gamma=1.5, n_bits=4).decoder.quantization_step == 0.1875
    assert BinarySAE(8, 16).n_bits == 8                                  # reference default
    assert float(b.encoder[0].bias.detach().abs().sum()) == 0.0
    assert float(b.encoder[0].weight.detach().abs().max()) <= (6 / (512 + 32768)) ** 0.5 * (1 + 1e-6)
    assert BaselineSparseAutoencoder(512, 1024).topk == 32
    t = TernarySparseAutoencoder(512, 32768)
    assert t.topk == 65 and t.decoder.threshold == 0.5
    q = QuantizedMatryoshkaSAE(512, 32768, top_k=32, abs_range=4, n_bits=4)
    assert q.decoder.nested_dictionary_size == [4096, 4096, 8192, 16384]
    assert q.decoder.quant_step == 0.5 and q.top_k == 32 and not q.decoder.needs_padding
    odd = QuantizedMatryoshkaSAE(64, 1000, top_k=32, n_bits=4)
    assert odd.decoder.nested_dictionary_size == [125, 125, 250, 500]
    assert odd.decoder.padded_sizes == [128, 128, 256, 512] and odd.decoder.needs_padding
    idx = odd.decoder.padded_index("cpu")
    assert idx.numel() == 1024 and (idx >= 0).sum() == 1000
    assert torch.equal(idx[idx >= 0], torch.arange(1000))
    base = SparseAutoencoder(4, 8)
    with pytest.raises(NotImplementedError):
        base.encode(torch.zeros(1, 4))
    with pytest.raises(NotImplementedError):
        base.decode(torch.zeros(1, 8))


def test_ternary_checkpoint_with_hook_buffers_loads():
    m = TernarySparseAutoencoder(16, 64)
    sd = dict(m.state_dict())
    sd["decoder.input_activations"] = torch.zeros(3, 64)     # what the reference saves after a forward
    m.load_state_dict(sd, strict=True)
    with pytest.raises(NotImplementedError):
        m.decoder.update_mask(0.1)


# ---- wrapper face ------------------------------------------------------------------------------------
def test_registry_and_loader_errors(tmp_path):
    assert list(F.SAE_REGISTRY)[:4] == ["b_sae", "q_sae", "rq_sae", "baseline_sae"]
    assert F.SAE_REGISTRY["b_sae"].kwargs == {"input_dim": 512, "hidden_dim": 32768, "gamma": 1.5, "n_bits": 4}
    assert F.SAE_REGISTRY["q_sae"].kwargs["abs_range"] == 1.5 and F.SAE_REGISTRY["q_sae"].kwargs["top_k"] == 32
    assert set(F.available_saes()) == set(F.SAE_REGISTRY)
    assert all(isinstance(p, Path) for p in F.available_saes().values())
    with pytest.raises(KeyError):
        F.load_sae("nope")
    with pytest.raises(FileNotFoundError):
        F.load_sae("b_sae")
    with pytest.raises(ValueError):
        F._ensure_tensor([])
    with pytest.raises(TypeError):
        F._ensure_tensor(np.zeros(3))
    t = torch.zeros(2)
    assert F._ensure_tensor((t, 1)) is t
    assert F._default_device("cpu") == torch.device("cpu")


def test_eleuther_safetensors_remap(tmp_path):
    from safetensors.torch import save_file
    H, D = 32, 8
    raw = {"encoder.weight": torch.randn(H, D), "encoder.bias": torch.randn(H), "W_dec": torch.randn(H, D),
           "b_dec": torch.randn(D)}
    p = tmp_path / "sae.safetensors"
    save_file(raw, str(p))
    import dataclasses
    entry = dataclasses.replace(F.SAE_REGISTRY["baseline_sae"], checkpoint_path=p, checkpoint_format="safetensors",
                                kwargs={"input_dim": D, "hidden_dim": H})
    sd = F._load_state_dict(entry)
    assert torch.equal(sd["decoder.weight"], raw["W_dec"].t()) and sd["decoder.weight"].shape == (D, H)
    BaselineSparseAutoencoder(D, H).load_state_dict(sd, strict=True)


def test_pure_python_safetensors_reader_and_loader_fallback(tmp_path, monkeypatch):
    """quantizedsae_amd/load_baseline.py (reference data/load_baseline.py:5-53) against files written by the real
    `safetensors` package, and the loader's branch for hosts without that package (framework.py:236-260)."""
    import dataclasses
    import json
    import struct
    from safetensors.torch import load_file, save_file
    from quantizedsae_amd.load_baseline import load_safetensors as local
    H, D = 48, 16
    g = torch.Generator().manual_seed(3)
    raw = {"encoder.weight": torch.randn(H, D, generator=g), "encoder.bias": torch.randn(H, generator=g),
           "W_dec": torch.randn(H, D, generator=g), "b_dec": torch.randn(D, generator=g),
           "half": torch.randn(5, 3, generator=g).half(), "bf": torch.randn(7, generator=g).bfloat16(),
           "i64": torch.arange(6).reshape(2, 3), "i32": torch.arange(4, dtype=torch.int32), "u8": torch.arange(9, dtype=torch.uint8),
           "flag": torch.tensor([True, False, True]), "scalar": torch.tensor(2.5), "empty": torch.zeros((0, 4))}
    path = tmp_path / "sae.safetensors"
    save_file(raw, str(path), metadata={"format": "pt"})
    got, want = local(path), load_file(str(path))
    assert set(got) == set(want) == set(raw)
    for k in raw:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    got["encoder.bias"] += 1.0                                            # the result owns writable memory
    # the loader without the safetensors package: same remapped state dict as with it
    entry = dataclasses.replace(F.SAE_REGISTRY["baseline_sae"], checkpoint_path=path, checkpoint_format="safetensors",
                                kwargs={"input_dim": D, "hidden_dim": H})
    with_pkg = F._load_state_dict(entry)
    monkeypatch.setattr(F, "load_safetensors", None)
    without = F._load_state_dict(entry)
    assert list(without) == list(with_pkg) == ["encoder.0.weight", "encoder.0.bias", "decoder.weight", "decoder.bias"]
    assert all(torch.equal(without[k], with_pkg[k]) for k in with_pkg)
    assert without["decoder.weight"].shape == (D, H) and without["decoder.weight"].is_contiguous()
    BaselineSparseAutoencoder(D, H).load_state_dict(without, strict=True)
    # malformed files are refused, not reinterpreted
    def write(hdr, payload=b"", n=None):
        h = json.dumps(hdr).encode()
        q = tmp_path / "bad.safetensors"
        q.write_bytes(struct.pack("<Q", len(h) if n is None else n) + h + payload)
        return q
    for bad in (write({"a": {"dtype": "F32", "shape": [2], "data_offsets": [0, 8]}}, b"\0" * 4),       # payload too short
                write({"a": {"dtype": "F32", "shape": [3], "data_offsets": [0, 8]}}, b"\0" * 8),       # shape x dtype != bytes
                write({"a": {"dtype": "F8_E4M3", "shape": [8], "data_offsets": [0, 8]}}, b"\0" * 8),   # unknown dtype
                write({"a": {"dtype": "F32", "shape": [2]}}, b"\0" * 8),                               # no offsets
                write({}, n=1 << 40)):                                                                 # absurd header length
        with pytest.raises(ValueError):
            local(bad)
    short = tmp_path / "short.safetensors"
    short.write_bytes(b"\x01\x02")
    with pytest.raises(ValueError):
        local(short)
    assert local(write({"__metadata__": {"k": "v"}})) == {}


# ---- sharding ------------------------------------------------------------------------------------------
def test_shard_rows_partitions_exactly():
    for n in [0, 1, 7, 64, 65536, 10_000_000, 10_000_001]:
        for ws in [1, 2, 3, 8]:
            spans = [sharding.shard_rows(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_rows(10, 2, 2)
    chunks = list(sharding.iter_chunk_shards([10_000_000, 5], 8, 3))
    assert chunks[0] == (0, 3_750_000, 5_000_000) and chunks[1] == (1, 3, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quantizedsae_amd import synthetic as S
    n_rows, D = 1001, 16
    x = S.normal(1, (n_rows, D), stream=1)
    recon = x + S.normal(1, (n_rows, D), stream=2, std=0.1)
    s, e = sharding.shard_rows(n_rows, world, rank)
    part = oracle.sq_err_sum(recon[s:e], x[s:e])                  # each rank reduces only its own rows
    mse = sharding.reduce_mse(torch.tensor(part, dtype=torch.float64), (e - s) * D)
    slow = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, mse, slow, (s, e)))
    dist.destroy_process_group()


def test_row_sharded_mse_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from quantizedsae_amd import synthetic as S
    x = S.normal(1, (1001, 16), stream=1)
    recon = x + S.normal(1, (1001, 16), stream=2, std=0.1)
    want = oracle.sq_err_sum(recon, x) / x.size
    assert res[0][3] == (0, 501) and res[1][3] == (501, 1001)
    for _, mse, slow, _ in res:
        assert mse == pytest.approx(want, rel=1e-12)
        assert slow == 2.0


def test_three_bf16_terms_carry_an_fp32_value_exactly():
    """The arithmetic claim behind csrc/split_dec_bf16.h, checked on the CPU with an emulated round-to-nearest-even bf16:
    t1 = bf16(v), t2 = bf16(v - t1), t3 = bf16(v - t1 - t2) -- both subtractions exact in fp32, t1 + t2 + t3 == v for every
    finite fp32 v whose third term stays normal (|v| >= 2^-100 here), so v * t (t in {-1, 0, +1}) reaches the accumulator of
    the bf16 matrix pipe without any rounding of its own."""
    def bf16(a):
        u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)

    rng = np.random.default_rng(0)
    mant = rng.integers(0, 1 << 23, 2_000_000, dtype=np.uint32)
    expo = rng.integers(27, 250, 2_000_000, dtype=np.uint32)             # 2^-100 .. 2^122
    sign = rng.integers(0, 2, 2_000_000, dtype=np.uint32)
    v = ((sign << 31) | (expo << 23) | mant).view(np.float32)
    v[:7] = [0.0, 1.0, -1.0, 3.0, np.float32(1) + np.float32(2 ** -23), 255.0, np.float32(2 ** -23)]
    t1 = bf16(v)
    r1 = (v - t1).astype(np.float32)
    assert np.array_equal(r1.astype(np.float64), v.astype(np.float64) - t1.astype(np.float64))      # exact subtraction
    t2 = bf16(r1)
    r2 = (r1 - t2).astype(np.float32)
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - t2.astype(np.float64))
    t3 = bf16(r2)
    assert np.array_equal(t3, r2)                                         # the last remainder needs no rounding
    total = t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64)
    assert np.array_equal(total, v.astype(np.float64))
    assert np.abs(t2[t1 != 0]).max() <= 2.0 ** -8 * np.abs(t1[t1 != 0]).max()
