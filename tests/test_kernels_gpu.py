"""Parity of every HIP kernel (through the C ABI, via quantizedsae_amd.ops) against the CPU
oracle on the same seeded inputs.  Bit-exact for the encoder chain, top-k indices, packers and
the sparse decode; 1e-5 relative for the dense decoders (tolerance of north_star)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from golden_util import rel_err
from quantizedsae_amd import synthetic as S

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from quantizedsae_amd import ops
    return ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


# ---- encoder ----------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [0])
@pytest.mark.parametrize("B,D,H", [(300, 64, 1000), (256, 512, 2048), (257, 48, 520), (1, 512, 33), (513, 32, 257),
                                   (1025, 512, 1280), (130, 96, 300)])
def test_encode_dense_bitexact(cfg, B, D, H):
    ops = _ops()
    x = S.activations(7, B, D)
    W = S.xavier_uniform(7, H, D, stream=1)
    bias = S.normal(7, (H,), stream=3, std=0.1)
    want = oracle.encode(x, W, bias, oracle.ACT_NONE)
    got = host(ops.encode_dense(dev(x), dev(W), dev(bias), ops.ACT_NONE))
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
    want_relu = np.maximum(want, 0)
    assert np.array_equal(host(ops.encode_dense(dev(x), dev(W), dev(bias), ops.ACT_RELU)), want_relu)
    got_nb = host(ops.encode_dense(dev(x), dev(W), None, ops.ACT_NONE))
    assert np.array_equal(got_nb, oracle.encode(x, W, None, oracle.ACT_NONE))
    sig = host(ops.encode_dense(dev(x), dev(W), dev(bias), ops.ACT_SIGMOID))
    np.testing.assert_allclose(sig, 1.0 / (1.0 + np.exp(-want.astype(np.float64))), rtol=2e-6, atol=1e-7)


def test_encode_dense_strided_out_and_empty():
    ops = _ops()
    x = S.activations(8, 70, 64)
    W = S.xavier_uniform(8, 96, 64, stream=1)
    big = torch.full((70, 128), -7.0, device=DEV)
    ops.encode_dense(dev(x), dev(W), None, ops.ACT_NONE, out=big[:, :96])
    assert np.array_equal(host(big[:, :96]), oracle.encode(x, W, None))
    assert (big[:, 96:] == -7.0).all()
    empty = ops.encode_dense(torch.empty((0, 64), device=DEV), dev(W), None)
    assert tuple(empty.shape) == (0, 96)


@pytest.mark.parametrize("cfg", [0])
@pytest.mark.parametrize("B,D,H", [(300, 64, 1000), (130, 512, 4096), (5, 48, 70)])
def test_encode_bits_bitexact(cfg, B, D, H):
    ops = _ops()
    x = S.activations(9, B, D)
    W = S.xavier_uniform(9, H, D, stream=1)
    bias = S.normal(9, (H,), stream=3, std=0.05)
    pre = oracle.encode(x, W, bias)
    want = oracle.zbits(pre)
    z = host(ops.encode_bits(dev(x), dev(W), dev(bias))).view(np.uint32)
    got = np.unpackbits(z.view(np.uint8), axis=1, bitorder="little")[:, :H]
    assert np.array_equal(got, want)
    # padding bits beyond H are zero
    assert np.unpackbits(z.view(np.uint8), axis=1, bitorder="little")[:, H:].sum() == 0


def test_encode_bits_cutoff_edge():
    """pre-activations sitting exactly on the fp32 sigmoid cutoff (x=1 on one input, W = value)."""
    ops = _ops()
    gt, _ = oracle.sigmoid_cutoffs()
    vals = np.array([gt, np.nextafter(gt, np.float32(-1)), 0.0, -0.0, 1e-7, 5e-8, -1e-9, 1.0], np.float32)
    H, D = 32, 4
    W = np.zeros((H, D), np.float32)
    W[: len(vals), 0] = vals
    x = np.zeros((3, D), np.float32)
    x[:, 0] = 1.0
    z = host(ops.encode_bits(dev(x), dev(W), None)).view(np.uint32)
    want = oracle.zbits(oracle.encode(x, W, None))
    got = np.unpackbits(z.view(np.uint8), axis=1, bitorder="little")[:, :H]
    assert np.array_equal(got, want)
    assert got[0, :8].tolist() == [1, 0, 0, 0, 1, 0, 0, 1]


# ---- top-k -------------------------------------------------------------------------------------
def _check_topk(lat, k):
    ops = _ops()
    want_idx, want_val = oracle.topk(lat, k)
    t = dev(lat)
    idx, val = ops.topk_rows(t, k, zero_rest=True)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    dense = host(t)
    want_dense = oracle.densify(want_idx, want_val, lat.shape[1])
    # NaN-safe comparison of the masked latent
    assert np.array_equal(np.nan_to_num(dense, nan=12345.0), np.nan_to_num(want_dense, nan=12345.0))
    t2 = dev(lat)
    idx2, _ = ops.topk_rows(t2, k, zero_rest=False)
    assert np.array_equal(host(idx2), want_idx)
    assert np.array_equal(host(t2).view(np.uint32), lat.view(np.uint32))


@pytest.mark.parametrize("H,k", [(32768, 65), (32768, 32), (32768, 1), (32768, 256), (1000, 2), (1000, 250),
                                 (2048, 65), (4, 3), (8192, 16), (16384, 65)])
def test_topk_random(H, k):
    lat = S.normal(3, (37, H), stream=H % 97)
    _check_topk(lat, k)


def test_topk_ties_and_specials():
    H = 32768
    rows = []
    rows.append(np.zeros(H, np.float32))                                   # all equal -> radix fallback
    rows.append(np.full(H, -3.5, np.float32))
    r = np.zeros(H, np.float32); r[::7] = 1.0; rows.append(r)               # massive ties at the boundary
    r = S.normal(5, (H,), stream=1); r[[5, 77, 30000]] = np.nan; rows.append(r)
    r = S.normal(5, (H,), stream=2); r[[1, 2]] = np.inf; r[[9, 10]] = -np.inf; rows.append(r)
    r = np.arange(H, dtype=np.float32); rows.append(r)                     # ascending
    rows.append(r[::-1].copy())                                            # descending
    r = S.normal(5, (H,), stream=3) * 1e-3                                  # large values owned by 10 threads only
    e = np.arange(H); own = ((e // 4) % 256) < 10
    r[own] = 5.0 + S.normal(5, (int(own.sum()),), stream=4); rows.append(r)
    r = np.round(S.normal(5, (H,), stream=6) * 2).astype(np.float32); rows.append(r)   # few distinct values
    r = np.where(np.arange(H) % 2 == 0, 0.0, -0.0).astype(np.float32); rows.append(r)  # +-0 ties
    lat = np.stack(rows)
    for k in (65, 1, 256):
        _check_topk(lat, k)


def test_topk_strided_rows():
    ops = _ops()
    lat = S.normal(6, (9, 1024), stream=1)
    buf = torch.zeros((9, 2048), device=DEV)
    buf[:, :1024] = dev(lat)
    idx, val = ops.topk_rows(buf[:, :1024], 5, zero_rest=True)
    want_idx, want_val = oracle.topk(lat, 5)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(buf[:, :1024]), oracle.densify(want_idx, want_val, 1024))


@pytest.mark.parametrize("B,D,H,k", [(70, 64, 1000, 2), (1500, 512, 2048, 65)])
def test_encode_topk_equals_two_step(B, D, H, k):
    ops = _ops()
    x = S.activations(10, B, D)
    W = S.xavier_uniform(10, H, D, stream=1)
    bias = S.normal(10, (H,), stream=3, std=0.05)
    idx, val = ops.encode_topk(dev(x), dev(W), dev(bias), k)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val), want_val)


# ---- BinarySAE dictionary ---------------------------------------------------------------------
@pytest.mark.parametrize("n_bits", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("D,H", [(512, 600), (48, 100), (20, 64)])
def test_pack_and_sparse_decode_binary(n_bits, D, H):
    ops = _ops()
    logits = S.normal(20 + n_bits, (H, D * n_bits), stream=1, std=3.0)
    gt, _ = oracle.sigmoid_cutoffs()
    logits[0, :8] = [0.0, -0.0, gt, np.nextafter(gt, np.float32(-1)), 1e-7, -1e-7, 5e-8, 30.0][: 8]
    packed, pol = ops.pack_binary(dev(logits), D, n_bits)
    want_packed = oracle.pack_binary(logits, D, n_bits)
    assert np.array_equal(host(packed), want_packed)
    total = H * D * n_bits
    assert float(pol.item()) / total == pytest.approx(oracle.polarize(logits, D, n_bits), rel=1e-5)
    assert np.array_equal(host(ops.unpack_binary(packed, D, n_bits)), oracle.unpack_binary(want_packed, D, n_bits))
    # sparse decode, several k including ragged tails of the unrolled loop
    for k in (1, 3, 4, 65):
        kk = min(k, H)
        B = 19
        lat = S.normal(30, (B, H), stream=k)
        idx, val = oracle.topk(lat, kk)
        bias = S.normal(31, (D,), stream=2, std=0.3)
        step = np.float32(4.0 / 2 ** (n_bits - 1))
        got = host(ops.decode_binary_sparse(dev(idx), dev(val), packed, D, n_bits, float(step), dev(bias)))
        want = oracle.decode_binary(idx, val, want_packed, D, n_bits, float(step), bias)
        assert np.array_equal(got, want), (n_bits, D, H, k)
    got = host(ops.decode_binary_sparse(dev(idx), dev(val), packed, D, n_bits, 0.1875, None))
    assert np.array_equal(got, oracle.decode_binary(idx, val, want_packed, D, n_bits, 0.1875, None))


def test_soft_table_and_table_decode():
    ops = _ops()
    H, D, n = 300, 64, 4
    logits = S.normal(40, (H, D * n), stream=1, std=2.0)
    table = host(ops.binary_soft_table(dev(logits), D, n))
    p = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    bw = np.array([1, 2, 4, -8], np.float64)
    want = (p.reshape(H, D, n) * bw).sum(-1)
    np.testing.assert_allclose(table, want, rtol=2e-6, atol=2e-6)
    lat = S.normal(41, (23, H), stream=1)
    idx, val = oracle.topk(lat, 32)
    tab = S.normal(42, (H, D), stream=2)
    bias = S.normal(42, (D,), stream=3)
    for scale in (1.0, 0.5):
        got = host(ops.decode_table_sparse(dev(idx), dev(val), dev(tab), scale, dev(bias)))
        assert np.array_equal(got, oracle.decode_table(idx, val, tab, scale, bias))


def test_densify_and_sq_err():
    ops = _ops()
    B, H, k = 33, 1000, 7
    lat = S.normal(50, (B, H), stream=1)
    idx, val = oracle.topk(lat, k)
    got = host(ops.densify(dev(idx), dev(val), H))
    assert np.array_equal(got, oracle.densify(idx, val, H))
    a = S.normal(51, (B, 515), stream=1)
    b = S.normal(51, (B, 515), stream=2)
    s = float(ops.sq_err_sum(dev(a), dev(b)).item())
    assert s == pytest.approx(oracle.sq_err_sum(a, b), rel=1e-12)


# ---- dense decoders ------------------------------------------------------------------------------
def _unpack2(words, n):
    w = words.view(np.uint32)
    f = (w[:, :, None] >> (2 * np.arange(16, dtype=np.uint32))[None, None, :]) & 3
    f = f.reshape(w.shape[0], -1)[:, :n].astype(np.int8)
    return np.where(f == 3, -1, f).astype(np.int8)


@pytest.mark.parametrize("B,D,H", [(100, 64, 1000), (300, 512, 4096), (7, 512, 32768)])
def test_ternary_pack_and_decode(B, D, H):
    ops = _ops()
    w = S.normal(60, (D, H), stream=1, std=0.5)
    w[0, :6] = [0.5, -0.5, 0.49999997, -0.49999997, 0.0, np.nan]
    codes = ops.pack_ternary(dev(w))
    want_codes = oracle.ternary_codes(w)
    assert np.array_equal(_unpack2(host(codes), H), want_codes)
    h = np.maximum(S.normal(61, (B, H), stream=2), 0).astype(np.float32)
    got = host(ops.decode_ternary_dense(dev(h), codes, D))
    want = oracle.decode_ternary(h, want_codes)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("B,D,H,n_bits,abs_range", [(100, 64, 1024, 4, 4.0), (300, 512, 4096, 4, 1.5),
                                                     (9, 512, 32768, 4, 4.0), (50, 64, 512, 1, 1.5)])
def test_matryoshka_pack_and_decode(B, D, H, n_bits, abs_range):
    ops = _ops()
    sd = S.matryoshka_sae_params(70, D, H, bias_std=0.3)
    w, wm, bias = sd["decoder.weight"], sd["decoder.weight_mirror"], sd["decoder.bias"]
    _, ge = oracle.sigmoid_cutoffs()
    w[0, :4] = [ge, np.nextafter(ge, np.float32(-1)), 0.0, -1e-7]
    codes, scale = ops.pack_matryoshka(dev(w), dev(wm), n_bits, abs_range)
    want_codes, want_scale = oracle.matryoshka_pack(w, wm, n_bits, abs_range)
    assert ops.matryoshka_sizes(H, n_bits) == oracle.matryoshka_sizes(H, n_bits)
    assert np.array_equal(_unpack2(host(codes), H).T * 2, want_codes)
    assert np.array_equal(host(scale), want_scale)
    zb = (S.uniform(71, (B, H), 0, 1, stream=3) < 0.01).astype(np.uint8)
    zwords = np.packbits(zb, axis=1, bitorder="little").view(np.int32)
    levels, counts = ops.decode_matryoshka(dev(zwords), H, D, n_bits, codes, scale, dev(bias), True)
    want_levels, want_l0 = oracle.decode_matryoshka(zb, want_codes, want_scale, bias, n_bits, True)
    for i in range(n_bits):
        assert rel_err(host(levels[i]), want_levels[i]) < 1e-5, i
    np.testing.assert_allclose(host(counts) / B, want_l0, rtol=1e-6)
    levels_nb, _ = ops.decode_matryoshka(dev(zwords), H, D, n_bits, codes, scale, dev(bias), False)
    want_nb, _ = oracle.decode_matryoshka(zb, want_codes, want_scale, bias, n_bits, False)
    assert rel_err(host(levels_nb[-1]), want_nb[-1]) < 1e-5


@pytest.mark.parametrize("B,D,H,shift", [(2500, 512, 8192, -2.5), (4096, 256, 4096, -2.0), (700, 128, 2048, -1.5),
                                         (2048, 512, 32768, -2.5), (1024, 512, 8192, 0.0)])
def test_encode_bits_prefilter_matches_exact(B, D, H, shift):
    """z bits from the fp16 candidate sweep + exact re-evaluation near the cutoff == the exact dense kernel's bits
    (which the oracle pins, test_encode_bits); shift = encoder bias in standard deviations of the latent
    (0: half of the units fire, every row overflows its list and takes the exact fallback)."""
    ops = _ops()
    x = S.activations(81, B, D)
    x[3] *= 40.0                                           # rows of very different norm: per-row scales and margins
    x[5] *= 1e-3
    W = S.xavier_uniform(81, H, D, stream=1)
    sigma = float(np.sqrt(D) * np.sqrt(6.0 / (D + H)) / np.sqrt(3.0))
    bias = (S.normal(81, (H,), stream=3, std=0.05 * sigma) + shift * sigma).astype(np.float32)
    # latents exactly at and around the cutoff: unit h0 has zero weights, its latent is its bias
    gt, _ = oracle.sigmoid_cutoffs()
    W[10] = 0.0; bias[10] = gt
    W[11] = 0.0; bias[11] = np.nextafter(gt, np.float32(-1))
    W[12] = 0.0; bias[12] = 0.0
    xd, Wd, bd = dev(x), dev(W), dev(bias)
    want = ops.encode_bits(xd, Wd, bd)
    Wq, meta = ops.prefilter_pack_w(Wd, bd)
    got, flagged = ops.encode_bits_prefilter(xd, Wd, bd, Wq, meta)
    assert torch.equal(got, want)
    if shift == 0.0:
        assert flagged == B
    elif shift <= -2.0:
        assert flagged < B // 8                            # (-1.5: ~7 % of the units fire, a quarter of the rows overflow)
    bits = np.unpackbits(host(got).view(np.uint8), axis=1, bitorder="little")
    assert bits[:, 10].all() and not bits[:, 11].any() and not bits[:, 12].any()
    # small oracle cross-check on a few rows
    lat = oracle.encode(x[:8], W, bias, oracle.ACT_NONE)
    assert np.array_equal(bits[:8, :H], (lat >= gt).astype(np.uint8))


def test_encode_bits_prefilter_nonfinite_rows():
    ops = _ops()
    B, D, H = 1024, 512, 4096
    x = S.activations(82, B, D)
    x[7, 3] = np.nan
    x[9, 100] = np.inf
    W = S.xavier_uniform(82, H, D, stream=1)
    bias = np.full((H,), -1.25, np.float32)                # ~ -2.6 standard deviations of the latent
    xd, Wd, bd = dev(x), dev(W), dev(bias)
    Wq, meta = ops.prefilter_pack_w(Wd, bd)
    got, flagged = ops.encode_bits_prefilter(xd, Wd, bd, Wq, meta)
    assert torch.equal(got, ops.encode_bits(xd, Wd, bd))
    assert 2 <= flagged < 64


@pytest.mark.parametrize("B,D,H,n_bits,density", [(300, 512, 4096, 4, 0.01), (1000, 512, 32768, 4, 0.006),
                                                   (257, 64, 1024, 4, 0.05), (130, 256, 2048, 3, 0.5),
                                                   (64, 1024, 4096, 8, 0.02), (33, 128, 512, 1, 0.1)])
def test_matryoshka_sparse_decode_matches_dense(B, D, H, n_bits, density):
    """the walk over the active units gives the dense kernel's levels bit for bit (same chain, zero terms skipped)"""
    ops = _ops()
    sd = S.matryoshka_sae_params(90, D, H, bias_std=0.3)
    w, wm, bias = dev(sd["decoder.weight"]), dev(sd["decoder.weight_mirror"]), dev(sd["decoder.bias"])
    sizes = [(s + 31) // 32 * 32 for s in ops.matryoshka_sizes(H, n_bits)]
    sizes[-1] = H - sum(sizes[:-1])
    assert sizes[-1] > 0 and sizes[-1] % 32 == 0
    codes, scale = ops.pack_matryoshka(w, wm, n_bits, 4.0, sizes)
    rows = ops.pack_matryoshka_rows(w, wm)
    zb = (S.uniform(91, (B, H), 0, 1, stream=3) < density).astype(np.uint8)
    zb[1] = 0                                              # an empty row
    zb[2] = 1                                              # every unit active
    zb[3] = 0; zb[3, H - 1] = 1                            # only the last unit
    zwords = dev(np.packbits(zb, axis=1, bitorder="little").view(np.int32))
    for allow_bias in (True, False):
        want, want_counts = ops.decode_matryoshka(zwords, H, D, n_bits, codes, scale, bias, allow_bias, sizes)
        got, counts = ops.decode_matryoshka_sparse(zwords, H, D, n_bits, rows, scale, bias, allow_bias, sizes)
        assert torch.equal(got, want)
        assert torch.equal(counts, want_counts)


def test_empty_batches_through_every_sparse_entry_point():
    """B = 0 is legal everywhere (the reference's modules accept an empty batch): empty outputs, no launch."""
    ops = _ops()
    D, H, k, n_bits = 512, 4096, 8, 4
    W = dev(S.xavier_uniform(5, H, D, stream=1))
    bias = torch.zeros((H,), device=DEV)
    x0 = torch.empty((0, D), device=DEV)
    Wq, meta = ops.prefilter_pack_w(W, bias)
    z, flagged = ops.encode_bits_prefilter(x0, W, bias, Wq, meta)
    assert tuple(z.shape) == (0, H // 32) and flagged == 0 and not ops.encode_bits_prefilter_supported(0, D, H)
    assert ops.encode_bits(x0, W, bias).shape == (0, H // 32)
    sd = S.matryoshka_sae_params(6, D, H)
    w, wm = dev(sd["decoder.weight"]), dev(sd["decoder.weight_mirror"])
    codes, scale = ops.pack_matryoshka(w, wm, n_bits, 4.0)
    rows = ops.pack_matryoshka_rows(w, wm)
    z0 = torch.empty((0, H // 32), dtype=torch.int32, device=DEV)
    for fn, c in ((ops.decode_matryoshka, codes), (ops.decode_matryoshka_sparse, rows)):
        levels, counts = fn(z0, H, D, n_bits, c, scale, None, False)
        assert tuple(levels.shape) == (n_bits, 0, D) and int(counts.sum()) == 0
    idx0 = torch.empty((0, k), dtype=torch.int32, device=DEV)
    val0 = torch.empty((0, k), device=DEV)
    assert int(ops.activation_counts(idx0, val0, H).sum()) == 0


def test_error_codes():
    from quantizedsae_amd import _lib
    ops = _ops()
    x = torch.zeros((4, 6), device=DEV)
    W = torch.zeros((8, 6), device=DEV)
    with pytest.raises(_lib.QsaeError) as e:
        ops.encode_dense(x, W, None)           # D % 4 != 0
    assert e.value.code == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.QsaeError):
        ops.topk_rows(torch.zeros((2, 64), device=DEV), 65, True)   # k > H
    with pytest.raises(RuntimeError):
        ops.encode_dense(torch.zeros((4, 8)), torch.zeros((8, 8)), None)   # CPU tensors: no fallback


# ---- fused encoder + top-k (pilot threshold, sweep filter, select, fallback) -----------------------
@pytest.fixture()
def fused_path():
    """Force the fused / prefilter pipelines on shapes far below their auto thresholds.  The switch exists in the debug
    build only (libqsae_hip_debug.so: same sources, -DQSAE_DEBUG_BUILD), so the test's calls are routed there for its
    duration; the product library has no such switch to flip."""
    from quantizedsae_amd import _lib
    with _lib.use_library("debug") as lib:
        lib.qsae_debug_set_topk_path.argtypes = [C.c_int]
        lib.qsae_debug_set_topk_path(2)
        try:
            yield lib
        finally:
            lib.qsae_debug_set_topk_path(0)


@pytest.mark.parametrize("B,D,H,k", [(300, 64, 4096, 8), (1000, 512, 8192, 65), (129, 48, 4100, 3), (2100, 512, 32768, 65)])
def test_encode_topk_fused_equals_oracle(fused_path, B, D, H, k):
    ops = _ops()
    x = S.activations(80, B, D)
    W = S.xavier_uniform(80, H, D, stream=1)
    bias = S.normal(80, (H,), stream=3, std=0.05)
    idx, val = ops.encode_topk(dev(x), dev(W), dev(bias), k)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    idx2, val2 = ops.encode_topk(dev(x), dev(W), None, k)
    w2 = oracle.topk(oracle.encode(x, W, None), k)
    assert np.array_equal(host(idx2), w2[0]) and np.array_equal(host(val2), w2[1])


def test_encode_topk_fused_fallback_rows(fused_path):
    """Rows the pilot threshold cannot serve: (a) all-equal latents -> every hidden unit ties, the
    candidate list overflows; (b) the whole top-k inside the pilot block -> fewer than k candidates;
    both must come back exact through the flagged-row fallback, next to ordinary rows."""
    ops = _ops()
    B, D, H, k = 260, 64, 4096, 40
    x = S.activations(81, B, D)
    W = S.xavier_uniform(81, H, D, stream=1)
    bias = np.zeros((H,), np.float32)
    x[3] = 0.0                                    # (a) latent row == bias == 0 everywhere
    x[200] = 0.0
    bias2 = bias.copy()
    bias2[:100] = 50.0                            # (b) top-40 of EVERY row sits in the first 100 hidden units
    for b_ in (bias, bias2):
        idx, val = ops.encode_topk(dev(x), dev(W), dev(b_), k)
        want_idx, want_val = oracle.topk(oracle.encode(x, W, b_), k)
        assert np.array_equal(host(idx), want_idx)
        assert np.array_equal(host(val), want_val)


# ---- K-interleaved operand layout ---------------------------------------------------------------------
def _kperm_host(a):
    r, K = a.shape
    g = a.reshape(r, K // 8, 8)
    return np.ascontiguousarray(g[:, :, [0, 2, 4, 6, 1, 3, 5, 7]].reshape(r, K))


@pytest.mark.parametrize("B,D,H", [(300, 64, 1000), (257, 512, 2048), (1, 32, 33)])
def test_kperm_rows_and_encode_dense_kperm(B, D, H):
    ops = _ops()
    x = S.activations(90, B, D)
    W = S.xavier_uniform(90, H, D, stream=1)
    bias = S.normal(90, (H,), stream=3, std=0.1)
    xp, Wp = ops.kperm_rows(dev(x)), ops.kperm_rows(dev(W))
    assert np.array_equal(host(xp), _kperm_host(x)) and np.array_equal(host(Wp), _kperm_host(W))
    want = oracle.encode(x, W, bias)
    assert np.array_equal(host(ops.encode_dense(xp, Wp, dev(bias), ops.ACT_NONE, kperm=True)), want)
    assert np.array_equal(host(ops.encode_dense(xp, Wp, dev(bias), ops.ACT_RELU, kperm=True)), np.maximum(want, 0))
    from quantizedsae_amd import _lib
    with pytest.raises(_lib.QsaeError):
        ops.kperm_rows(torch.zeros((4, 12), device=DEV))          # K % 8 != 0


@pytest.mark.parametrize("B,D,H,k", [(1000, 512, 8192, 65), (300, 64, 4096, 8)])
def test_encode_topk_kperm_fused_and_chunked(fused_path, B, D, H, k):
    ops = _ops()
    lib = fused_path
    x = S.activations(91, B, D)
    W = S.xavier_uniform(91, H, D, stream=1)
    bias = S.normal(91, (H,), stream=3, std=0.05)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    xp, Wp = ops.kperm_rows(dev(x)), ops.kperm_rows(dev(W))
    for path in (2, 1):
        lib.qsae_debug_set_topk_path(path)
        idx, val = ops.encode_topk(xp, Wp, dev(bias), k, kperm=True)
        assert np.array_equal(host(idx), want_idx) and np.array_equal(host(val), want_val)


@pytest.mark.parametrize("B,D,H,k,kperm", [(1000, 512, 8192, 65, True), (300, 64, 4096, 8, False), (70, 64, 1000, 2, False)])
def test_encode_topk_latent_dense_output(fused_path, B, D, H, k, kperm):
    """encoder + top-k + the reference's dense `latent * mask` in one call (zero-fill fused in the sweep)."""
    ops = _ops()
    x = S.activations(92, B, D)
    W = S.xavier_uniform(92, H, D, stream=1)
    bias = S.normal(92, (H,), stream=3, std=0.05)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    a_x, a_W = (ops.kperm_rows(dev(x)), ops.kperm_rows(dev(W))) if kperm else (dev(x), dev(W))
    torch.cuda.synchronize()
    poison = torch.full((B, H), 7.0, device=DEV)            # make sure every element gets written
    del poison
    idx, val, dense = ops.encode_topk_latent(a_x, a_W, dev(bias), k, kperm=kperm)
    assert np.array_equal(host(idx), want_idx) and np.array_equal(host(val), want_val)
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))


# ---- fp16 prefilter: exact results through an approximate candidate pass ---------------------------------
def _prefilter(ops, x, W, bias, k, want_dense=True, info=None, spec_rows=0):
    Wq, meta = ops.prefilter_pack_w(dev(W), dev(bias) if bias is not None else None)
    return ops.encode_topk_prefilter(dev(x), dev(W), dev(bias) if bias is not None else None, Wq, meta, k,
                                     want_dense=want_dense, info=info, spec_rows=spec_rows)


@pytest.mark.parametrize("B,D,H,k", [(1000, 512, 8192, 65), (300, 64, 4096, 8), (2100, 512, 32768, 65)])
def test_prefilter_equals_oracle(fused_path, B, D, H, k):
    ops = _ops()
    x = S.activations(95, B, D)
    W = S.xavier_uniform(95, H, D, stream=1)
    bias = S.normal(95, (H,), stream=3, std=0.05)
    assert ops.prefilter_supported(B, D, H, k)
    idx, val, dense = _prefilter(ops, x, W, bias, k)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))
    idx2, val2, none = _prefilter(ops, x, W, None, k, want_dense=False)
    w2 = oracle.topk(oracle.encode(x, W, None), k)
    assert none is None and np.array_equal(host(idx2), w2[0]) and np.array_equal(host(val2), w2[1])


@pytest.mark.parametrize("B,D,H,k", [(515, 256, 8192, 16), (700, 128, 4096, 8), (257, 512, 4096, 65), (40, 128, 4096, 8)])
def test_prefilter_stationary_sweep_shapes(fused_path, B, D, H, k):
    """The activation-stationary sweep (D in {128, 256, 512}) on ragged batches, biased and unbiased."""
    ops = _ops()
    x = S.activations(98, B, D)
    W = S.xavier_uniform(98, H, D, stream=1)
    bias = S.normal(98, (H,), stream=3, std=0.1)
    assert ops.prefilter_supported(B, D, H, k)
    idx, val, dense = _prefilter(ops, x, W, bias, k)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))


def test_prefilter_stationary_sweep_strided_dense_and_degenerate_rows(fused_path):
    """D = 512 (the stationary sweep): dense latent into a strided buffer whose padding must stay untouched;
    rows that tie everywhere, hold inf / NaN, or flood their candidate slots take the exact fallback."""
    ops = _ops()
    B, D, H, k = 530, 512, 8192, 65
    x = S.activations(99, B, D)
    W = S.xavier_uniform(99, H, D, stream=1)
    bias = S.normal(99, (H,), stream=3, std=0.02)
    x[3] = 0.0                                             # latent == bias: dense ties at the top are possible
    x[10, 5] = np.inf
    x[11, 7] = np.nan
    x[20] *= 1e-30                                         # latents ~ bias: thousands of near-equal candidates
    Wq, meta = ops.prefilter_pack_w(dev(W), dev(bias))
    buf = torch.full((B, H + 64), 7.0, dtype=torch.float32, device=DEV)
    idx, val, dense = ops.encode_topk_prefilter(dev(x), dev(W), dev(bias), Wq, meta, k, dense_out=buf)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    ok = np.ones(B, bool); ok[[10, 11]] = False
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val)[ok].view(np.uint32), want_val[ok].view(np.uint32))
    assert np.array_equal(host(dense)[ok], oracle.densify(want_idx, want_val, H)[ok])
    assert bool((buf[:, H:] == 7.0).all())                 # the fill respects the row stride


def test_prefilter_ordered_checkpoint_keeps_the_fast_path(fused_path):
    """Hidden units ordered by liveness (the first H/8 never fire): the stratified in-kernel pilot still finds a
    useful threshold -- exact results and (nearly) no row in the exact fallback."""
    from quantizedsae_amd import _lib
    ops = _ops()
    B, D, H, k = 600, 512, 8192, 16                        # k = int(H * 0.002), the reference's ratio
    x = S.activations(101, B, D)
    W = S.xavier_uniform(101, H, D, stream=1)
    bias = S.normal(101, (H,), stream=3, std=0.02)
    bias[: H // 8] = -50.0                                 # dead units first
    info = {}
    idx, val, dense = _prefilter(ops, x, W, bias, k, info=info)
    flagged = info["flagged_rows"]
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))
    assert flagged <= B // 20, flagged


def test_prefilter_error_bound_holds_with_margin(fused_path):
    """max |approx - exact chain| over the pilot block stays below the eps_b the selection relies on (inputs with
    outliers, tiny and huge scales).  eps_b is built from the MEASURED distances between the operands and their fp16
    copies (Cauchy-Schwarz on them), so the headroom is the slack of that inequality: ~2x or more on hardware."""
    from quantizedsae_amd import _lib
    ops = _ops()
    B, D, H, k = 600, 512, 8192, 65
    x = S.activations(96, B, D)
    x[::7] *= 1e-6
    x[1::7] *= 3e4
    x[2::7, ::5] = 0.0
    x[3::7, 3] = 250.0                                     # single outlier feature
    W = S.xavier_uniform(96, H, D, stream=1)
    W[::3] *= 0.01
    bias = S.normal(96, (H,), stream=3, std=0.5)
    lib = fused_path                                       # the debug build (see the fixture)
    lib.qsae_debug_set_inkernel_pilot.argtypes = [C.c_int, C.c_int]
    lib.qsae_debug_set_inkernel_pilot(0, 0)                # this test reads the dense pilot block: separate pilot GEMM
    try:
        idx, val, _ = _prefilter(ops, x, W, bias, k, want_dense=False)
    finally:
        lib.qsae_debug_set_inkernel_pilot(1, 0)             # (rank 0 = derived from k)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx) and np.array_equal(host(val), want_val)
    po, mo, pc = C.c_size_t(), C.c_size_t(), C.c_int()
    lib.qsae_debug_prefilter_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 2 + [C.POINTER(C.c_int)]
    lib.qsae_debug_prefilter_offsets(B, D, H, k, C.byref(po), C.byref(mo), C.byref(pc))
    ws = ops._workspace(torch.device(DEV), 1)             # this stream's scratch buffer, as the call above left it
    P = pc.value
    approx = host(ws[po.value: po.value + B * P * 4].view(torch.float32).reshape(B, P))
    margin = host(ws[mo.value: mo.value + B * 4].view(torch.float32))
    exact = oracle.encode(x, W[:P], bias[:P])
    ratio = np.abs(approx.astype(np.float64) - exact).max(axis=1) / (margin / 2.0)
    assert np.isfinite(ratio).all()
    assert ratio.max() < 0.6, ratio.max()                  # the bound itself is ratio <= 1


def test_candidate_lists_of_the_sweep_respect_the_error_budget(fused_path):
    """What the candidate sweep leaves in the lists (in-kernel pilot, bias folded into the MFMA chain, records scaled back
    at flush time): every entry's value is within eps_b of the exact chain of its hidden unit, every hidden unit whose
    exact latent reaches tau - margin + eps_b is listed, nothing is listed twice."""
    ops = _ops()
    B, D, H, k = 520, 512, 8192, 65
    x = S.activations(99, B, D)
    x[::5] *= 30.0
    x[1::5] *= 1e-3
    W = S.xavier_uniform(99, H, D, stream=1)
    bias = S.normal(99, (H,), stream=3, std=0.3)
    idx, val, _ = _prefilter(ops, x, W, bias, k, want_dense=False)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx) and np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    lib = fused_path
    offs = [C.c_size_t() for _ in range(5)]
    cap, parts = C.c_int(), C.c_int()
    lib.qsae_debug_prefilter_list_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 5 + [C.POINTER(C.c_int)] * 2
    lib.qsae_debug_prefilter_list_offsets(B, D, H, k, *[C.byref(o) for o in offs], C.byref(cap), C.byref(parts))
    cand_off, cnt_off, cntp_off, tau_off, mar_off = (o.value for o in offs)
    cap, parts = cap.value, parts.value
    ws = ops._workspace(torch.device(DEV), 1)             # this stream's scratch buffer, as the call above left it
    cand = host(ws[cand_off: cand_off + B * cap * 8].view(torch.int32).reshape(B, cap, 2))
    cnt0 = host(ws[cnt_off: cnt_off + B * 4].view(torch.int32))
    cntp = host(ws[cntp_off: cntp_off + B * 4 * 7].view(torch.int32).reshape(7, B))
    tau = host(ws[tau_off: tau_off + B * 4].view(torch.float32))
    margin = host(ws[mar_off: mar_off + B * 4].view(torch.float32))
    exact = oracle.encode(x, W, bias)
    cap_part = cap // parts
    worst = 0.0
    for b in range(B):
        hs, vs = [], []
        for p in range(parts):
            n = int(cnt0[b] if p == 0 else cntp[p - 1][b])
            if n > cap_part:                               # overflowing rows are flagged: not this test's subject
                hs = None
                break
            seg = cand[b, p * cap_part: p * cap_part + n]
            vs.append(seg[:, 0].copy().view(np.float32))
            hs.append(seg[:, 1])
        if hs is None:
            continue
        hs, vs = np.concatenate(hs), np.concatenate(vs)
        assert len(set(hs.tolist())) == len(hs) and hs.min(initial=0) >= 0 and hs.max(initial=0) < H
        eps = margin[b] / 2.0
        err = np.abs(vs.astype(np.float64) - exact[b, hs]).max(initial=0.0)
        worst = max(worst, err / eps)
        assert err <= eps, (b, err, eps)
        must = np.nonzero(exact[b] >= tau[b] - margin[b] + eps)[0]
        assert np.isin(must, hs).all(), b
    assert worst > 0.0


def test_refinement_hands_corrupted_lists_to_the_exact_kernels(fused_path):
    """The refinement gathers with what the candidate lists say.  Lists that could not have come from the sweep -- a hidden
    unit listed twice (two survivors with one key: a winner slot would stay unwritten and the row decode would gather
    with it), an index outside the dictionary -- make it flag the row instead; the exact kernels then produce the row.
    The debug build can run the two halves of a call separately, so the lists are edited in between."""
    ops = _ops()
    lib = fused_path
    lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
    B, D, H, k = 520, 512, 8192, 65
    x = S.activations(89, B, D)
    W = S.xavier_uniform(89, H, D, stream=1)
    bias = S.normal(89, (H,), stream=3, std=0.1)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    offs = [C.c_size_t() for _ in range(5)]
    cap, parts = C.c_int(), C.c_int()
    lib.qsae_debug_prefilter_list_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 5 + [C.POINTER(C.c_int)] * 2
    lib.qsae_debug_prefilter_list_offsets(B, D, H, k, *[C.byref(o) for o in offs], C.byref(cap), C.byref(parts))
    cand_off, cnt_off = offs[0].value, offs[1].value
    try:
        lib.qsae_debug_set_phases(1, 0)                    # activation preparation + candidate sweep only
        _prefilter(ops, x, W, bias, k, want_dense=False)
        torch.cuda.synchronize()
        ws = ops._workspace(torch.device(DEV), 1)
        cand = ws[cand_off: cand_off + B * cap.value * 8].view(torch.int32).reshape(B, cap.value, 2)
        cnt0 = host(ws[cnt_off: cnt_off + B * 4].view(torch.int32))
        assert (cnt0[:12] >= 2).all()
        # rows 0-9: the row's best hidden unit (a certain survivor) written over the first two entries: listed twice or
        # three times, with its exact value as the approximate one
        top = torch.from_numpy(np.stack([want_val[:10, 0].view(np.int32), want_idx[:10, 0].astype(np.int32)], axis=1)).to(cand.device)
        cand[0:10, 0, :] = top
        cand[0:10, 1, :] = top
        cand[10, 0, 1] = H + 5                             # row 10: a hidden index past the dictionary
        cand[11, 0, 1] = -1                                # row 11: 0xFFFFFFFF
        info = {}
        lib.qsae_debug_set_phases(2, 0)                    # refinement (+ exact fallback) on the edited lists
        idx, val, _ = _prefilter(ops, x, W, bias, k, want_dense=False, info=info)
    finally:
        lib.qsae_debug_set_phases(3, 0)
    assert info["flagged_rows"] >= 12
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))


def test_prefilter_degenerate_rows_fall_back(fused_path):
    ops = _ops()
    B, D, H, k = 260, 64, 4096, 40
    x = S.activations(97, B, D)
    W = S.xavier_uniform(97, H, D, stream=1)
    bias = np.zeros((H,), np.float32)
    x[3] = 0.0                                             # every latent ties
    x[10, 5] = np.inf                                      # non-finite rows
    x[11, 7] = np.nan
    idx, val, dense = _prefilter(ops, x, W, bias, k)
    lat = oracle.encode(x, W, bias)
    want_idx, want_val = oracle.topk(lat, k)
    ok = np.ones(B, bool); ok[[10, 11]] = False            # inf/NaN rows: compare index sets only where finite
    assert np.array_equal(host(idx)[ok], want_idx[ok])
    assert np.array_equal(host(val)[ok], want_val[ok])
    assert np.array_equal(host(idx)[~ok], want_idx[~ok])   # exact path ranks NaN/inf rows like the oracle
    # with spec_rows the exact pass for the first 32 flagged rows runs speculatively (count read on the device while
    # the host waits for it): same results
    idx2, val2, dense2 = _prefilter(ops, x, W, bias, k, spec_rows=32)
    assert torch.equal(idx2, idx) and torch.equal(val2.view(torch.int32), val.view(torch.int32))
    if dense is not None:
        assert torch.equal(dense2.view(torch.int32), dense.view(torch.int32))


def test_prefilter_with_biases_beyond_the_scaled_range(fused_path):
    """The sweep starts its MFMA chains from bias * s_x s_w (a power of two around 2^17 for these inputs): a bias that
    overflows there cannot be represented in the candidate pass.  The error budget declares such rows unservable, so they
    take the exact kernels -- and the outputs stay the oracle's, with the huge units on top."""
    ops = _ops()
    B, D, H, k = 300, 128, 4096, 16
    x = S.activations(98, B, D)
    W = S.xavier_uniform(98, H, D, stream=1)
    bias = S.normal(98, (H,), stream=3, std=0.05)
    bias[7] = 3.0e35
    bias[11] = -3.0e35
    info = {}
    idx, val, dense = _prefilter(ops, x, W, bias, k, info=info)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))
    assert (host(idx)[:, 0] == 7).all() and info["flagged_rows"] == B


def test_ablation_builds_of_the_encoder_run_clean_at_the_shape_that_once_faulted():
    """Round 1, 18:15-18:20: `tools/bench_ablate.py` (B=32768, D=512, H=16384) ended twice in a GPU memory-access fault
    (gpurun_out/abl3.log: address 0x3bf3daf5c000; abl4.log: address nil).  Cause: the ablated
    instantiations (no loads / no LDS traffic inside the K loop) had been built on top of the then-new inline-asm staging
    loads, whose completion only the complete pipeline waits for; the two prologue load sets were still in flight when the
    registers were reused, and a late 16-byte load landed in a register pair that by then held an address.  Since 15326ec
    ablated builds use compiler-visible loads.  This runs the three builds once at that shape in the debug library (the
    product library has neither the entry point nor the ablated kernels); build 0 must equal the product encoder."""
    from quantizedsae_amd import _lib
    ops = _ops()
    B, D, H = 32768, 512, 16384
    g = torch.Generator(device=DEV); g.manual_seed(3)
    x = torch.randn((B, D), device=DEV, generator=g)
    W = (torch.rand((H, D), device=DEV, generator=g) * 2 - 1) * 0.0134
    want = ops.encode_dense(x, W, None)
    out = torch.empty((B, H), device=DEV)
    with _lib.use_library("debug") as lib:
        f = lib.qsae_debug_encode_ablate
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for ablate in (1, 2, 0):
            assert f(x.data_ptr(), W.data_ptr(), B, D, H, out.data_ptr(), 2, ablate, stream) == 0
            torch.cuda.synchronize()
    assert torch.equal(out, want)                              # the last build run (0) is the complete pipeline


# ---- refinement as three launches, chains slice-major (large batches by default; forced here on small shapes) ----------
@pytest.fixture
def sliced_refinement(fused_path):
    lib = fused_path
    lib.qsae_debug_set_refine_sliced.argtypes = [C.c_int]
    lib.qsae_debug_set_refine_sliced(2)
    try:
        yield lib
    finally:
        lib.qsae_debug_set_refine_sliced(1)


@pytest.mark.parametrize("B,D,H,k", [(1000, 512, 8192, 65), (300, 64, 4096, 8), (2100, 512, 32768, 65), (515, 256, 16384, 32),
                                     (700, 1024, 8192, 64), (129, 64, 4100, 3), (640, 512, 8192, 130), (1030, 512, 32768, 20)])
def test_sliced_refinement_equals_oracle_and_the_one_launch_form(sliced_refinement, B, D, H, k):
    """select / slice-major chains / rank == the oracle's top-k, and bit for bit the one-launch refinement; rows not a
    multiple of the 128-row wave tasks, one to many slices (H D 4 / 4 MiB rounded to eight), k on both sides of 64; with
    k = 20 over 16 slices a row has 1-2 survivors per slice, so a batch of 64 pairs spans more activation rows than the chain
    launch's tile holds (24) and is cut short."""
    ops = _ops()
    lib = sliced_refinement
    x = S.activations(195, B, D)
    W = S.xavier_uniform(195, H, D, stream=1)
    bias = S.normal(195, (H,), stream=3, std=0.05)
    assert ops.prefilter_supported(B, D, H, k)
    info = {}
    idx, val, dense = _prefilter(ops, x, W, bias, k, info=info)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))
    assert np.array_equal(host(dense), oracle.densify(want_idx, want_val, H))
    lib.qsae_debug_set_refine_sliced(0)
    info0 = {}
    idx0, val0, dense0 = _prefilter(ops, x, W, bias, k, info=info0)
    lib.qsae_debug_set_refine_sliced(2)
    assert torch.equal(idx, idx0) and torch.equal(val.view(torch.int32), val0.view(torch.int32)) and torch.equal(dense, dense0)
    assert info["flagged_rows"] == info0["flagged_rows"]
    idx2, val2, none = _prefilter(ops, x, W, None, k, want_dense=False)
    w2 = oracle.topk(oracle.encode(x, W, None), k)
    assert none is None and np.array_equal(host(idx2), w2[0]) and np.array_equal(host(val2), w2[1])


@pytest.mark.parametrize("n_bits,D", [(4, 512), (8, 512), (2, 512), (0, 512), (4, 256), (4, 1024), (8, 128), (3, 256)])
def test_sliced_refinement_decodes_like_the_one_launch_form(sliced_refinement, n_bits, D):
    """The rank launch's row decode -- packed 4-bit fields (the 13-instruction nibble form), 8-bit fields, narrower fields
    (generic form), an fp32 table (n_bits 0) -- against the one-launch refinement and the stand-alone decode kernel;
    degenerate rows (all ties, NaN, inf) take the exact kernels in both forms."""
    ops = _ops()
    lib = sliced_refinement
    # (dictionary rows of 32 and 128 dwords as well as the usual 64: the wide decode hands list entries between lanes with
    # v_readlane, so every lane has to run every round of its loop -- a row narrower than the wave once read garbage there)
    B, H, k = 1100, 8192, 65
    x = S.activations(196, B, D)
    x[5] = 0.0
    x[17, 3] = np.nan
    x[40, 100] = np.inf
    W = S.xavier_uniform(196, H, D, stream=1)
    bias = S.normal(196, (H,), stream=3, std=0.05)
    dbias = dev(S.normal(196, (D,), stream=5, std=0.1))
    xd, Wd, bd = dev(x), dev(W), dev(bias)
    Wq, meta = ops.prefilter_pack_w(Wd, bd)
    if n_bits:
        logits = dev(S.normal(196, (H, D * n_bits), stream=7, std=30.0))
        packed, _ = ops.pack_binary(logits, D, n_bits)
        call = lambda: ops.binary_forward_prefilter(xd, Wd, bd, Wq, meta, k, packed, n_bits, 0.25, dbias)
    else:
        table = dev(S.normal(196, (H, D), stream=7, std=1.0))
        call = lambda: ops.table_forward_prefilter(xd, Wd, bd, Wq, meta, k, table, 0.5, dbias)
    got = call()
    lib.qsae_debug_set_refine_sliced(0)
    want = call()
    lib.qsae_debug_set_refine_sliced(2)
    for a, b_ in zip(got, want):
        assert torch.equal(a.view(torch.int32), b_.view(torch.int32))
    if n_bits:
        rec = ops.decode_binary_sparse(got[0], got[1], packed, D, n_bits, 0.25, dbias)
        ok = np.ones(B, bool); ok[[17, 40]] = False
        assert np.array_equal(host(rec)[ok], host(got[3])[ok])


def test_sliced_refinement_sorts_lists_that_are_not_in_slice_order(sliced_refinement):
    """The select launch stores the survivors as they come when their slices already form ascending runs (the sweep's order)
    and sorts them otherwise; a unit listed twice or an index outside the dictionary still hands the row to the exact
    kernels.  The lists are edited between the two halves of a call (debug build)."""
    ops = _ops()
    lib = sliced_refinement
    lib.qsae_debug_set_phases.argtypes = [C.c_int, C.c_int]
    B, D, H, k = 520, 512, 32768, 65                      # 16 slices of 2048 units
    x = S.activations(197, B, D)
    W = S.xavier_uniform(197, H, D, stream=1)
    bias = S.normal(197, (H,), stream=3, std=0.1)
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    offs = [C.c_size_t() for _ in range(5)]
    cap, parts = C.c_int(), C.c_int()
    lib.qsae_debug_prefilter_list_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 5 + [C.POINTER(C.c_int)] * 2
    lib.qsae_debug_prefilter_list_offsets(B, D, H, k, *[C.byref(o) for o in offs], C.byref(cap), C.byref(parts))
    cand_off, cnt_off, cntp_off = offs[0].value, offs[1].value, offs[2].value
    P, cap_part = parts.value, cap.value // parts.value   # small batches: the hidden range in P parts, one list segment each
    try:
        lib.qsae_debug_set_phases(1, 0)
        _prefilter(ops, x, W, bias, k, want_dense=False)
        torch.cuda.synchronize()
        ws = ops._workspace(torch.device(DEV), 1)
        cand = ws[cand_off: cand_off + B * cap.value * 8].view(torch.int32).reshape(B, cap.value, 2)
        cnt = np.zeros((P, B), np.int64)
        cnt[0] = host(ws[cnt_off: cnt_off + B * 4].view(torch.int32))
        if P > 1:
            cnt[1:] = host(ws[cntp_off: cntp_off + (P - 1) * B * 4].view(torch.int32)).reshape(P - 1, B)
        assert (cnt.sum(0) >= k).all() and (cnt <= cap_part).all() and (cnt[0, 300:311] >= 2).all()
        g = torch.Generator().manual_seed(5)
        for b in range(0, 300):                            # rows 0-199: every segment back to front (descending slices inside
            for p_ in range(P):                            # a part); rows 200-299: every segment in a random order
                n, lo = int(cnt[p_, b]), p_ * cap_part
                order = torch.arange(n - 1, -1, -1) if b < 200 else torch.randperm(n, generator=g)
                cand[b, lo:lo + n] = cand[b, lo:lo + n][order.to(cand.device)]
        top = torch.from_numpy(np.stack([want_val[300:310, 0].view(np.int32), want_idx[300:310, 0].astype(np.int32)], axis=1)).to(cand.device)
        cand[300:310, 0, :] = top                          # rows 300-309: the best unit listed twice
        cand[300:310, 1, :] = top
        cand[310, 0, 1] = H + 5
        info = {}
        lib.qsae_debug_set_phases(2, 0)
        idx, val, _ = _prefilter(ops, x, W, bias, k, want_dense=False, info=info)
    finally:
        lib.qsae_debug_set_phases(3, 0)
    assert 11 <= info["flagged_rows"] < 40
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32))


def test_sliced_refinement_ranks_equal_latents_like_the_oracle(sliced_refinement):
    """Hidden units with identical weights give identical exact latents in every row: the rank launch first ranks on the
    32-bit value keys, notices the tie (the ranks no longer sum to m (m - 1) / 2) and redoes the walk on the 64-bit keys, whose
    low half is the oracle's tie-break (lower index first).  Pairs and a triple, scaled so that they sit in most rows' top-k."""
    ops = _ops()
    lib = sliced_refinement
    B, D, H, k = 700, 512, 8192, 65
    x = S.activations(198, B, D)
    W = S.xavier_uniform(198, H, D, stream=1)
    bias = S.normal(198, (H,), stream=3, std=0.05)
    for src, dsts in ((3, (10,)), (4000, (500,)), (77, (78, 8000))):
        W[src] *= 3.0
        for d in dsts:
            W[d] = W[src]
            bias[d] = bias[src]
    want_idx, want_val = oracle.topk(oracle.encode(x, W, bias), k)
    tied_rows = sum(len(np.unique(want_val[b])) < k for b in range(B))
    assert tied_rows > B // 4                              # the case is exercised
    for form in (2, 0):
        lib.qsae_debug_set_refine_sliced(form)
        idx, val, dense = _prefilter(ops, x, W, bias, k)
        assert np.array_equal(host(idx), want_idx), form
        assert np.array_equal(host(val).view(np.uint32), want_val.view(np.uint32)), form
    lib.qsae_debug_set_refine_sliced(2)
