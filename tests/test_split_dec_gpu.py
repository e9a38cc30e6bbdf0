"""The dense decoders on the bf16 matrix pipe (csrc/split_dec_bf16.h): qsae_decode_ternary_dense_split and
qsae_decode_matryoshka_split against the exact-fp32 kernels and the fp64-accumulating oracle.

Reference arithmetic: sae/ternary.py:41-52 (recon = h @ hard^T, hard in {-1, 0, +1}), sae/quantized_matryoshka.py:67-129
(recon_i = sum over levels <= i of (scale z) @ S, S in {-2, 0, 2}, + bias).  Contract: the fp32 operand is split EXACTLY into
three bf16 terms, every product is exact, accumulation is fp32 -- so (a) wherever all partial sums are exactly representable
(small-integer activations) the result is bit-identical to the fp32 kernel's and the oracle's, and (b) on real-valued
activations it stays at the fp32 kernel's distance from the fp64 oracle (both far inside the 1e-5 bar of north_star)."""
import numpy as np
import pytest
import torch

import oracle
from quantizedsae_amd import ops, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 512


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def rel_err(got, want):
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(np.abs(want).max(), 1e-30))


@pytest.mark.parametrize("B,H", [(300, 4096), (1024, 32768), (129, 64), (4096, 8192)])
def test_ternary_split_against_fp32_kernel_and_oracle(B, H):
    w = S.normal(160, (D, H), stream=1, std=0.5)
    codes = ops.pack_ternary(dev(w))
    want_codes = oracle.ternary_codes(w)
    tq = ops.expand_codes_bf16(codes, D, H)
    assert ops.split_dec_supported(B, H, D)
    # ReLU latents over a wide dynamic range: rows scaled by 1e-6 .. 1e4, exact zeros, a few huge entries
    h = np.maximum(S.normal(161, (B, H), stream=2), 0).astype(np.float32)
    h *= (10.0 ** S.uniform(162, (B, 1), -6, 4, stream=3)).astype(np.float32)
    h[::7, ::13] *= 1000.0
    hd = dev(h)
    got = host(ops.decode_ternary_dense_split(hd, tq, D))
    f32 = host(ops.decode_ternary_dense(hd, codes, D))
    want = oracle.decode_ternary(h, want_codes)                     # fp64 accumulation
    # per row: the rows differ in scale by ten orders of magnitude
    for rows in (slice(0, B),):
        scale = np.abs(want[rows]).max(axis=1, keepdims=True) + 1e-30
        e_split = np.abs(got[rows].astype(np.float64) - want[rows]).max(axis=1) / scale[:, 0]
        e_f32 = np.abs(f32[rows].astype(np.float64) - want[rows]).max(axis=1) / scale[:, 0]
        assert e_split.max() < 1e-5 and e_f32.max() < 1e-5
        assert e_split.max() < 4.0 * max(e_f32.max(), 2e-7)     # same size as the fp32 kernel's accumulation rounding
    assert rel_err(got, want) < 1e-5


def test_ternary_split_is_exact_where_the_sums_are():
    """Integer activations below 2^8: three-term split, products and every partial sum are exact integers < 2^24, whatever
    the order -- split kernel == fp32 kernel == oracle, bit for bit.  Also: a strided (ld > H) activation matrix."""
    B, H = 260, 4096
    w = S.normal(170, (D, H), stream=1, std=0.5)
    codes = ops.pack_ternary(dev(w))
    tq = ops.expand_codes_bf16(codes, D, H)
    h = np.floor(S.uniform(171, (B, H), 0, 200, stream=2)).astype(np.float32)
    want = oracle.decode_ternary(h, oracle.ternary_codes(w))
    got = ops.decode_ternary_dense_split(dev(h), tq, D)
    assert np.array_equal(host(got), want)
    assert torch.equal(got, ops.decode_ternary_dense(dev(h), codes, D))
    wide = torch.zeros((B, H + 64), dtype=torch.float32, device=DEV)
    wide[:, :H] = dev(h)
    assert torch.equal(ops.decode_ternary_dense_split(wide[:, :H], tq, D), got)
    # fractional values with 24 significant bits: the split loses none of them (one non-zero column per row)
    hv = np.zeros((64, H), np.float32)
    vals = (1.0 + S.uniform01(172, 64)).astype(np.float32) * np.float32(2.0) ** np.arange(-40, 24).astype(np.float32)
    cols = np.argmax(oracle.ternary_codes(w)[0] != 0)               # a unit with a non-zero code in output column 0
    hv[np.arange(64), cols] = vals
    one = host(ops.decode_ternary_dense_split(dev(hv), tq, D))
    assert np.array_equal(one, oracle.decode_ternary(hv, oracle.ternary_codes(w)))


@pytest.mark.parametrize("B,H,n_bits,abs_range,density", [(300, 4096, 4, 1.5, 0.3), (1024, 32768, 4, 4.0, 0.5),
                                                          (130, 2048, 1, 1.5, 0.5), (2048, 8192, 4, 4.0, 0.02)])
def test_matryoshka_split_against_fp32_kernel_and_oracle(B, H, n_bits, abs_range, density):
    sd = S.matryoshka_sae_params(180, D, H, bias_std=0.3)
    w, wm, bias = sd["decoder.weight"], sd["decoder.weight_mirror"], sd["decoder.bias"]
    sizes = ops.matryoshka_sizes(H, n_bits)
    assert all(sum(sizes[:i + 1]) % 64 == 0 for i in range(n_bits))
    codes, scale = ops.pack_matryoshka(dev(w), dev(wm), n_bits, abs_range)
    want_codes, want_scale = oracle.matryoshka_pack(w, wm, n_bits, abs_range)
    tq, s3 = ops.expand_codes_bf16(codes, D, H), ops.split_scale_bf16(scale)
    # the three terms carry 2 * scale exactly
    assert np.array_equal(host(s3.float().sum(0)), 2.0 * want_scale)
    zb = (S.uniform(181, (B, H), 0, 1, stream=3) < density).astype(np.uint8)
    zwords = dev(np.packbits(zb, axis=1, bitorder="little").view(np.int32))
    for allow_bias in (True, False):
        levels, counts = ops.decode_matryoshka_split(zwords, H, D, n_bits, tq, s3, dev(bias), allow_bias)
        f32, counts32 = ops.decode_matryoshka(zwords, H, D, n_bits, codes, scale, dev(bias), allow_bias)
        want, want_l0 = oracle.decode_matryoshka(zb, want_codes, want_scale, bias, n_bits, allow_bias)
        assert torch.equal(counts, counts32)
        np.testing.assert_allclose(host(counts) / B, want_l0, rtol=1e-6)
        for i in range(n_bits):
            e_split, e_f32 = rel_err(host(levels[i]), want[i]), rel_err(host(f32[i]), want[i])
            assert e_split < 1e-5 and e_f32 < 1e-5, (i, e_split, e_f32)
            assert e_split < 4.0 * max(e_f32, 2e-7), (i, e_split, e_f32)


def test_matryoshka_split_with_an_empty_level_and_unsupported_shapes():
    B, H, n_bits = 200, 4096, 4
    sd = S.matryoshka_sae_params(190, D, H)
    w, wm = dev(sd["decoder.weight"]), dev(sd["decoder.weight_mirror"])
    sizes = [1024, 0, 1024, 2048]                              # level 1 is empty: it repeats level 0's output
    codes, scale = ops.pack_matryoshka(w, wm, n_bits, 4.0, sizes)
    tq, s3 = ops.expand_codes_bf16(codes, D, H), ops.split_scale_bf16(scale)
    zb = (S.uniform(191, (B, H), 0, 1, stream=3) < 0.2).astype(np.uint8)
    zwords = dev(np.packbits(zb, axis=1, bitorder="little").view(np.int32))
    levels, _ = ops.decode_matryoshka_split(zwords, H, D, n_bits, tq, s3, None, True, sizes)
    f32, _ = ops.decode_matryoshka(zwords, H, D, n_bits, codes, scale, None, True, sizes)
    assert torch.equal(levels[1], levels[0])
    for i in range(n_bits):
        assert rel_err(host(levels[i]), host(f32[i]).astype(np.float64)) < 2e-6
    assert not ops.split_dec_supported(B, H, 256) and not ops.split_dec_supported(B, 4096 + 32, D)
    with pytest.raises(ValueError):
        ops.expand_codes_bf16(codes, D, H + 32)
    from quantizedsae_amd._lib import QsaeError
    with pytest.raises(QsaeError, match="multiples of 64"):
        ops.decode_matryoshka_split(zwords, H, D, n_bits, tq, s3, None, True, [1024 + 32, 1024 - 32, 1024, 1024])


def test_models_take_the_split_decoders_by_default_and_fp32_on_request():
    from quantizedsae_amd import QuantizedMatryoshkaSAE, TernarySparseAutoencoder
    x = dev(S.activations(200, 512, D))
    tern = TernarySparseAutoencoder(D, 4096).to(DEV).eval()
    with torch.no_grad():
        tern.decoder.weight.normal_(0, 0.5)
    assert tern.decoder.resolved_precision(512) == "split"
    h, rec = tern(x)
    tern.decoder.precision = "fp32"
    h32, rec32 = tern(x)
    assert torch.equal(h, h32) and rel_err(host(rec), host(rec32).astype(np.float64)) < 2e-6 and not torch.equal(rec, rec32)
    small = TernarySparseAutoencoder(64, 1024).to(DEV).eval()     # input_dim != 512: the fp32 kernel, silently
    assert small.decoder.resolved_precision(16) == "fp32"
    small(dev(S.activations(201, 16, 64)))
    mat = QuantizedMatryoshkaSAE(D, 4096, top_k=32, abs_range=4, n_bits=4).to(DEV).eval()      # random init: dense regime
    mat.bits_path = "dense"
    g, lv = mat(x)
    assert "tq" in mat.decoder.packed()
    mat.decoder.precision = "fp32"
    g32, lv32 = mat(x)
    assert all(torch.equal(a, b) for a, b in zip(g, g32))
    # (half of the units fire at random init: sums of ~2000 cancelling terms; the two kernels round them in different orders)
    assert all(rel_err(host(a), host(b).astype(np.float64)) < 1e-5 for a, b in zip(lv, lv32))


# ---- dense-regime z bits: fp16 classification of every latent + exact resolution of the uncertainty band --------------------
@pytest.mark.parametrize("B,Dm,H,shift", [(2500, 512, 8192, 0.0), (4096, 256, 4096, -1.0), (700, 128, 2048, 0.5),
                                          (2048, 512, 32768, 0.0), (300, 512, 4128, -2.5)])
def test_encode_bits_band_matches_exact(B, Dm, H, shift):
    """qsae_encode_bits_band == qsae_encode_bits (the exact fp32 contraction; reference: encoder(x) then `latent > 0.5`,
    sae/quantized_matryoshka.py:97-99,206-209) bit for bit, at activation densities from 0.6 % to 70 %, with rows the bound
    cannot serve (NaN / inf) and rows of very different scale."""
    sd = S.matryoshka_sae_params(300, Dm, H, enc_bias_sigmas=shift)
    W, b = dev(sd["encoder.0.weight"]), dev(sd["encoder.0.bias"])
    x = S.activations(301, B, Dm)
    x[::9] *= 50.0
    x[1::9] *= 1e-3
    x[5, 3] = np.nan
    x[77, 0] = np.inf
    xd = dev(x)
    assert ops.encode_bits_band_supported(B, Dm, H)
    Wq, meta = ops.prefilter_pack_w(W, b)
    z, flagged = ops.encode_bits_band(xd, W, b, Wq, meta)
    want = ops.encode_bits(xd, W, b)
    assert torch.equal(z, want)
    assert 2 <= flagged <= 2 + B // 50                      # the two non-finite rows (+ rows whose band overflows: none expected)
    dens = float(torch.ops.qsae.encode_bits(xd, W, b).view(torch.uint8).to(torch.int32).sum()) if False else None


# ---- fp32-accurate encoder on the fp16 matrix pipe (opt-in) -----------------------------------------------------------------
@pytest.mark.parametrize("B,Dm,H,act", [(300, 512, 4096, 1), (1024, 512, 32768, 0), (257, 128, 1000, 2), (64, 64, 256, 1)])
def test_emulated_encoder_is_fp32_accurate(B, Dm, H, act):
    """qsae_encode_dense_emu against the exact fp32 chain (qsae_encode_dense) and an fp64 contraction: two fp16 terms per
    operand, three partial contractions, every product exact.  It is as far from the fp64 result as the fp32 chain is (a few
    1e-7 of the latent scale), and within 2e-6 of the chain; rows of very different scale and a non-finite row included."""
    W = S.xavier_uniform(400, H, Dm, stream=1)
    b = S.normal(400, (H,), stream=3, std=0.1)
    x = S.activations(401, B, Dm)
    x[::5] *= 300.0
    x[1::5] *= 1e-4
    Wd, bd, xd = dev(W), dev(b), dev(x)
    Wc, meta2 = ops.emu_pack_w(Wd)
    got = host(ops.encode_dense_emu(xd, Wc, meta2, bd, act))
    f32 = host(ops.encode_dense(xd, Wd, bd, act))
    pre = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    want = np.maximum(pre, 0) if act == 1 else (1 / (1 + np.exp(-pre)) if act == 2 else pre)
    scale = np.abs(pre).max(axis=1, keepdims=True) + 1e-30            # per row: the rows differ by six orders of magnitude
    e_emu = (np.abs(got - want) / scale).max()
    e_f32 = (np.abs(f32 - want) / scale).max()
    # measured without the row scaling (tools/experiments/r03_emu_err.py): emulated rms 0.7-0.9e-7 / max 1.1e-6 of the row's
    # largest latent, exact fp32 chain rms 1.1e-7 / max 1.3-1.8e-6 -- the emulation is at least as close to the real numbers
    assert e_emu < 4e-6 and e_f32 < 4e-6 and e_emu < 1.5 * e_f32 + 2e-7, (e_emu, e_f32)
    assert (np.abs(got.astype(np.float64) - f32) / scale).max() < 4e-6
    xb = x.copy()
    xb[3, 7] = np.nan
    bad = host(ops.encode_dense_emu(dev(xb), Wc, meta2, bd, act))
    assert np.isnan(bad[3]).all() and np.array_equal(bad[4], got[4])


def test_ternary_model_with_the_emulated_encoder():
    from quantizedsae_amd import TernarySparseAutoencoder
    sd = S.ternary_sae_params(410, D, 8192)
    model = TernarySparseAutoencoder(D, 8192)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(DEV).eval()
    x = S.activations(411, 1000, D)
    h0, r0 = model(dev(x))
    want = oracle.ternary_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"])
    assert np.array_equal(host(h0), want["latent"])                    # default: the exact chain, bit for bit
    model.encoder.precision = "emulated"
    h1, r1 = model(dev(x))
    assert rel_err(host(h1), want["latent"]) < 4e-6 and rel_err(host(r1), want["reconstruction"]) < 1e-5
    assert not torch.equal(h1, h0)
