"""End-to-end parity of the drop-in module classes and the SAEWrapper on the GPU:
HIP path vs CPU oracle (bit-exact where the contract says so) and vs the golden vectors the
reference produced (index sets outside audited near-ties, reconstructions within 1e-5)."""
import dataclasses

import numpy as np
import pytest
import torch

import oracle
from golden_util import Fixture, NEAR_TIE_EPS, rel_err, residual_clear_rows, row_rel_err
from quantizedsae_amd import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,
                              TernarySparseAutoencoder, synthetic as S)
from quantizedsae_amd.inference import framework as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RECON_TOL = 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def load(model, sd):
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return model.to(DEV).eval()


def audit_topk_sets(fx, idx_sorted):
    want = fx["topk_idx"]
    same = (idx_sorted == want).all(axis=1)
    bad = ~same & (fx["gap"] > NEAR_TIE_EPS)
    assert not bad.any(), f"{fx.name}: rows {np.nonzero(bad)[0][:8]} differ, gaps {fx['gap'][bad][:8]}"
    return same


@pytest.mark.parametrize("name", ["binary_small", "binary_n8", "binary_n2", "binary_mid", "binary_full_g4",
                                  "binary_full_g15"])
def test_binary_sae(name):
    fx = Fixture(name)
    m = fx.meta
    sd = fx.state_dict()
    rows = m["rows"]
    x = fx.x()[:rows]
    model = load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
    model.k = m["k"] / m["H"]
    assert model.top_k == m["k"]
    latent, recon, pol = model(dev(x))
    latent, recon = host(latent), host(recon)
    # --- bit-exact against the oracle -------------------------------------------------------
    want = oracle.binary_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                 sd["decoder.bias"], n_bits=m["n_bits"], gamma=m["gamma"], k=m["k"])
    assert np.array_equal(latent, want["latent"])
    assert np.array_equal(recon, want["reconstruction"])
    assert float(pol) == pytest.approx(want["polarize_loss"], rel=1e-5, abs=1e-20)
    # --- against the reference's golden outputs ---------------------------------------------
    nz = latent != 0
    assert (nz.sum(1) == m["k"]).all()
    idx_sorted = np.stack([np.nonzero(r)[0] for r in nz]).astype(np.int32)
    same = audit_topk_sets(fx, idx_sorted)
    assert row_rel_err(recon, fx["reconstruction"])[same].max() < RECON_TOL
    assert float(pol) == pytest.approx(float(fx["polarize_loss"]), rel=2e-5, abs=1e-18)
    # compact path returns the same selection and reconstruction
    idx, val, recon_c = model.forward_compact(dev(x))
    assert np.array_equal(host(idx), want["idx"]) and np.array_equal(host(val), want["val"])
    assert np.array_equal(host(recon_c), recon)


@pytest.mark.parametrize("name", ["binary_soft_small", "binary_soft_init", "binary_soft_n8", "binary_soft_mid"])
def test_binary_sae_unpolarised_checkpoints_follow_the_reference_forward(name):
    """Checkpoints whose decoder logits are not saturated (N(0, s^2) logits, the reference's default init): the
    reference forward multiplies with the soft sigmoid-bit integers (sae/binary.py:24-47).  decode_mode='auto' (the
    default) notices at pack time (soft_gap) and reproduces the reference's reconstructions to 1e-5; the explicit
    'hard' mode is the packed two's-complement decode and is far from them on such a checkpoint."""
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    rows = m["rows"]
    x = fx.x()[:rows]
    model = load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
    model.k = m["k"] / m["H"]
    assert model.decoder.decode_mode == "auto"
    with pytest.warns(UserWarning, match="not polarised"):
        latent, recon, pol = model(dev(x))
    assert model.decoder.resolved_decode_mode() == "soft"
    assert model.decoder.packed()["soft_gap"] == pytest.approx(float(fx["soft_gap"]), rel=1e-4)
    latent, recon = host(latent), host(recon)
    want = oracle.binary_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                 sd["decoder.bias"], n_bits=m["n_bits"], gamma=m["gamma"], k=m["k"], soft=True)
    assert np.array_equal(latent, want["latent"])                       # the encoder side is exact as ever
    assert rel_err(recon, want["reconstruction"]) < RECON_TOL           # (expf of the device vs the host's)
    nz = latent != 0
    idx_sorted = np.stack([np.nonzero(r)[0] for r in nz]).astype(np.int32)
    same = audit_topk_sets(fx, idx_sorted)
    assert row_rel_err(recon, fx["reconstruction"])[same].max() < RECON_TOL
    assert float(pol) == pytest.approx(float(fx["polarize_loss"]), rel=2e-5)
    if "soft_int_weights" in fx:
        assert np.max(np.abs(host(model.decoder.quantized_int_weights_continuous()) - fx["soft_int_weights"])) \
            < 2.0 ** (m["n_bits"] - 22)
        assert np.array_equal(host(model.decoder.quantized_int_weights()), fx["int_weights"].astype(np.float32))
        # the dense entry point of the decoder (reference signature) follows the same choice
        rec_d, _ = model.decoder(dev(fx["sparse_latent"]), None)
        assert rel_err(host(rec_d), fx["reconstruction"]) < RECON_TOL
    # compact path and explicit modes
    idx, val, recon_c = model.forward_compact(dev(x))
    assert np.array_equal(host(recon_c), recon)
    model.decoder.decode_mode = "soft"
    assert np.array_equal(host(model(dev(x))[1]), recon)
    model.decoder.decode_mode = "hard"
    hard = oracle.binary_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                 sd["decoder.bias"], n_bits=m["n_bits"], gamma=m["gamma"], k=m["k"])
    rec_h = host(model(dev(x))[1])
    assert np.array_equal(rec_h, hard["reconstruction"])
    assert row_rel_err(rec_h, fx["reconstruction"])[same].max() > 1e-2


def test_binary_auto_mode_takes_the_packed_decode_on_polarised_checkpoints():
    """+-30 logits: soft_gap ~1e-12, auto == hard (bit for bit), no warning; +-12 logits: gap 1e-4 -> soft."""
    import warnings
    fx = Fixture("binary_small")
    m, sd = fx.meta, fx.state_dict()
    model = load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, rec_auto, _ = model(dev(fx.x()))
    assert model.decoder.resolved_decode_mode() == "hard" and model.decoder.packed()["soft_gap"] < 1e-9
    model.decoder.decode_mode = "hard"
    assert torch.equal(model(dev(fx.x()))[1], rec_auto)
    model.decoder.decode_mode = "auto"
    with torch.no_grad():
        model.decoder.weight.mul_(12.0 / 30.0)                  # version bump: repacked, gap re-measured
    with pytest.warns(UserWarning, match="not polarised"):
        model(dev(fx.x()))
    assert 1e-5 < model.decoder.packed()["soft_gap"] < 1e-3 and model.decoder.resolved_decode_mode() == "soft"


def test_binary_auto_mode_measures_saturation_per_bit_not_per_integer():
    """The soft / hard gap grows with the integer range: an 8-bit dictionary at +-18 has gap 255 sigmoid(-18) = 3.9e-6 --
    above an absolute 1e-6, yet every bit is as saturated as a 4-bit dictionary's at the same logits (gap 2.3e-7).  The
    auto rule compares gap / (2^n_bits - 1) with hard_max_gap, so both take the packed decode; +-12 takes neither."""
    import warnings
    for n_bits in (4, 8):
        m = BinarySAE(64, 1024, gamma=4.0, n_bits=n_bits).to(DEV).eval()
        g = torch.Generator(device="cpu").manual_seed(n_bits)
        bits = (torch.rand(m.decoder.weight.shape, generator=g) > 0.5).to(DEV)
        with torch.no_grad():
            m.decoder.weight.copy_(torch.where(bits, 18.0, -18.0))
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            assert m.decoder.resolved_decode_mode() == "hard"
        gap = m.decoder.packed()["soft_gap"]
        assert gap == pytest.approx((2 ** n_bits - 1) * 1.523e-8, rel=0.3) and gap <= m.decoder.hard_gap_limit()
        with torch.no_grad():
            m.decoder.weight.mul_(12.0 / 18.0)
        with pytest.warns(UserWarning, match="not polarised"):
            assert m.decoder.resolved_decode_mode() == "soft"


def test_binary_k_zero_and_limits():
    """hidden_dim < 500 -> k = int(H * 0.002) = 0: the reference's topk(0) keeps nothing (zero latent, bias-only
    reconstruction, sae/binary.py:94-99); limits of the kernels are reported as ValueError, not as a HIP error."""
    m = BinarySAE(64, 400, gamma=4.0, n_bits=4).to(DEV).eval()
    with torch.no_grad():
        m.decoder.bias.copy_(torch.arange(64, device=DEV) * 0.25)
        m.decoder.weight.copy_(torch.where(torch.rand_like(m.decoder.weight) > 0.5, 30.0, -30.0))
    assert m.top_k == 0
    x = torch.randn(5, 64, device=DEV)
    lat, rec, pol = m(x)
    assert lat.shape == (5, 400) and not lat.any()
    assert torch.equal(rec, m.decoder.bias.detach().expand(5, -1))
    idx, val, rec_c = m.forward_compact(x)
    assert idx.shape == (5, 0) and torch.equal(rec_c, rec)
    m2 = BinarySAE(64, 1024, gamma=4.0, n_bits=4).to(DEV).eval()
    m2.k = 0.3                                                  # 307 > 256
    with pytest.raises(ValueError, match="limit of 256"):
        m2(torch.randn(4, 64, device=DEV))


def test_invalidate_packed_after_data_edit():
    """In-place edits through .data bump no version counter (the reference's own ternary code edits weights that way):
    invalidate_packed() is the documented way to make the model repack."""
    m = BinarySAE(64, 1024, gamma=4.0, n_bits=4).to(DEV).eval()
    m.decoder.decode_mode = "hard"
    x = torch.randn(8, 64, device=DEV)
    m.decoder.weight.data.copy_(torch.where(torch.rand_like(m.decoder.weight) > 0.5, 30.0, -30.0))
    m.invalidate_packed()
    r1 = m(x)[1].clone()
    m.decoder.weight.data.mul_(-1.0)                            # flips every bit, invisibly to the cache
    assert torch.equal(m(x)[1], r1)                             # ... which therefore still decodes the old dictionary
    m.invalidate_packed()
    assert not torch.equal(m(x)[1], r1)


def test_binary_decoder_dense_entry_and_exports():
    fx = Fixture("binary_small")
    m, sd = fx.meta, fx.state_dict()
    model = load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
    int_w = host(model.decoder.quantized_int_weights())
    assert np.array_equal(int_w, fx["int_weights"].astype(np.float32))
    # decoder(latent, x): arbitrary dense latent through the same dictionary
    lat = fx["sparse_latent"]
    recon, pol = model.decoder(dev(lat), None)
    assert rel_err(host(recon), fx["reconstruction"]) < RECON_TOL
    soft = host(model.decoder.quantized_int_weights_continuous())
    assert np.max(np.abs(soft - int_w)) < 1e-5            # saturated logits: soft == hard
    model.decoder.decode_mode = "soft"
    _, recon_soft, _ = model(dev(fx.x()))
    assert rel_err(host(recon_soft), fx["reconstruction"]) < RECON_TOL
    model.decoder.decode_mode = "bogus"
    with pytest.raises(ValueError):
        model(dev(fx.x()))


def test_binary_repack_after_weight_update():
    m = BinarySAE(64, 512, gamma=4.0, n_bits=4).to(DEV).eval()
    x = torch.randn(8, 64, device=DEV)
    with torch.no_grad():
        m.decoder.weight.copy_(torch.where(torch.rand_like(m.decoder.weight) > 0.5, 30.0, -30.0))
    r1 = m(x)[1].clone()
    with torch.no_grad():
        m.decoder.weight.mul_(-1.0)          # flips every bit: cache must notice the version bump
    r2 = m(x)[1]
    assert not torch.equal(r1, r2)


@pytest.mark.parametrize("name", ["baseline_small", "baseline_mid", "baseline_full"])
def test_baseline_sae(name):
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    x = fx.x()[: m["rows"]]
    model = load(BaselineSparseAutoencoder(m["D"], m["H"]), sd)
    h, recon = model(dev(x))
    h, recon = host(h), host(recon)
    want = oracle.baseline_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                   sd["decoder.bias"], k=m["k"])
    assert np.array_equal(h, want["latent"])
    assert np.array_equal(recon, want["reconstruction"])
    nz = h != 0
    idx_sorted = np.stack([np.nonzero(r)[0] for r in nz]).astype(np.int32)
    same = audit_topk_sets(fx, idx_sorted)
    assert row_rel_err(recon, fx["reconstruction"])[same].max() < RECON_TOL
    dense = host(model.apply_topk_activation(dev(want["latent_full"])))
    assert np.array_equal(dense, want["latent"])


@pytest.mark.parametrize("name", ["ternary_small", "ternary_mid", "ternary_full"])
def test_ternary_sae(name):
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    model = load(TernarySparseAutoencoder(m["D"], m["H"]), sd)
    h, recon = model(dev(fx.x()))
    want = oracle.ternary_forward(fx.x(), sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"])
    assert np.array_equal(host(h), want["latent"])                       # exact fp32 chain + ReLU
    assert rel_err(host(recon), want["reconstruction"]) < RECON_TOL
    assert rel_err(host(recon), fx["reconstruction"]) < RECON_TOL
    rows = fx["latent"].shape[0]                                         # (the full-size fixture stores the first rows + digests)
    assert np.max(np.abs(host(h)[:rows] - fx["latent"])) < 4e-6
    if "latent_sum" in fx:
        np.testing.assert_allclose(host(h).astype(np.float64).sum(1), fx["latent_sum"], rtol=1e-6)


@pytest.mark.parametrize("B,H,n_bits", [(2304, 8192, 4), (4096, 32768, 4), (2100, 8192, 8), (2048, 8192, 2)])
def test_binary_forward_fused_decode_matches_separate_kernels(B, H, n_bits):
    """BinarySAE.forward with the decode inside the refinement kernel (qsae_binary_forward_prefilter) == prefilter
    + stand-alone decode kernel, bit for bit, including rows that take the exact fallback (NaN / inf / all-ties)."""
    D = 512
    sd = S.binary_sae_params(77, D, H, n_bits, 30.0, 0.05, 0.1)
    model = load(BinarySAE(D, H, gamma=4.0, n_bits=n_bits), sd)
    x = S.activations(78, B, D)
    x[5] = 0.0
    x[17, 3] = np.nan
    x[40, 100] = np.inf
    xd = dev(x)
    assert model.resolved_latent_path(B) == "prefilter"
    outs = {}
    for fuse in (False, True, True):                      # the second fused call runs the speculative fallback
        model.fuse_decode = fuse
        lat, rec, pol = model(xd)
        idx, val, rec_c = model.forward_compact(xd)
        if fuse not in outs:
            outs[fuse] = (lat, rec, idx, val, rec_c)
        else:
            for a, b_ in zip(outs[fuse], (lat, rec, idx, val, rec_c)):
                assert torch.equal(a.view(torch.int32), b_.view(torch.int32))
    for a, b_ in zip(outs[False], outs[True]):
        assert torch.equal(a.view(torch.int32), b_.view(torch.int32))
    ok = np.ones(B, bool); ok[[17, 40]] = False
    want = oracle.binary_forward(x[:64], sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                 sd["decoder.bias"], n_bits=n_bits, gamma=4.0, k=model.top_k)
    assert np.array_equal(host(outs[True][1])[:64][ok[:64]], want["reconstruction"][ok[:64]])


@pytest.mark.parametrize("B,shift", [(4096, -2.5), (2304, -2.0), (2048, 0.0)])
def test_matryoshka_prefilter_path_matches_dense_path(B, shift):
    """QuantizedMatryoshkaSAE.forward through the candidate sweep + sparse walk == the exact dense kernels, bit for
    bit; with half of the units firing (shift 0) the model notices and stays on the dense kernels."""
    D, H, n_bits = 512, 32768, 4
    sd = S.matryoshka_sae_params(95, D, H, enc_bias_sigmas=shift, bias_std=0.2)
    model = load(QuantizedMatryoshkaSAE(D, H, 32, abs_range=4, n_bits=n_bits), sd)
    x = dev(S.activations(96, B, D))
    model.bits_path = "dense"
    gs, ls = model(x)                                     # dense decoder on the bf16 matrix pipe (default where it applies)
    model.decoder.precision = "fp32"                      # the exact-fp32 chain: what the sparse walk reproduces bit for bit
    g0, l0 = model(x)
    for i in range(n_bits):                               # same sums, another rounding order
        assert float(gs[i]) == float(g0[i]) and rel_err(host(ls[i]), host(l0[i])) < RECON_TOL
    model.bits_path = "auto"
    assert model.resolved_bits_path(B) == "prefilter"
    g1, l1 = model(x)
    for i in range(n_bits):
        assert torch.equal(l0[i], l1[i]), i
        assert float(g0[i]) == float(g1[i])
    if shift == 0.0:
        # dense activations: the model leaves the candidate lists for the band classification (same bits, same levels)
        assert model.last_flagged_rows == B and model.resolved_bits_path(B) == "band"
        g2, l2 = model(x)
        assert model.last_flagged_rows == 0
        for i in range(n_bits):
            assert torch.equal(l0[i], l2[i]) and float(g0[i]) == float(g2[i])
    else:
        assert model.last_flagged_rows < B // 8 and model.resolved_bits_path(B) == "prefilter"
        # a CPU cross-check of the first rows against the oracle
        want = oracle.matryoshka_forward(host(x[:16]), sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                         sd["decoder.weight_mirror"], sd["decoder.bias"], n_bits=n_bits, abs_range=4)
        for i in range(n_bits):
            assert rel_err(host(l1[i][:16]), want["reconstruction_levels"][i]) < RECON_TOL, i


@pytest.mark.parametrize("name", ["matryoshka_small", "matryoshka_edge", "matryoshka_mid", "matryoshka_full"])
def test_matryoshka_sae(name):
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    model = load(QuantizedMatryoshkaSAE(m["D"], m["H"], 32, abs_range=m["abs_range"], n_bits=m["n_bits"]), sd)
    assert model.decoder.nested_dictionary_size == m["sizes"]
    x = fx.x()
    groups, levels = model(dev(x))
    assert len(groups) == m["n_bits"] and len(levels) == m["n_bits"]
    want = oracle.matryoshka_forward(x, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                     sd["decoder.weight_mirror"], sd["decoder.bias"], n_bits=m["n_bits"],
                                     abs_range=m["abs_range"])
    np.testing.assert_allclose([float(g) for g in groups], want["latent_groups"], rtol=1e-6)
    np.testing.assert_allclose([float(g) for g in groups], fx["latent_groups"], rtol=1e-6)   # exact bit counts
    for i in range(m["n_bits"]):
        assert rel_err(host(levels[i]), want["reconstruction_levels"][i]) < RECON_TOL, i
        assert rel_err(host(levels[i]), fx["reconstruction_levels"][i]) < RECON_TOL, i
    # the bits themselves, mapped back from the padded hidden order
    zb = host(model.activation_bits(dev(x))).view(np.uint32)
    bits = np.unpackbits(zb.view(np.uint8), axis=1, bitorder="little")
    st = model.decoder.packed()
    if st["index"] is not None:
        index = host(st["index"])
        bits = bits[:, : len(index)][:, index >= 0]
    bits = bits[:, : m["H"]]
    assert np.array_equal(bits, np.unpackbits(fx["zbits"], axis=1)[:, : m["H"]])
    # decoder called directly with the dense sigmoid latent (as the reference's analysis scripts do)
    lat = model.encoder(dev(x))
    g2, l2 = model.decoder(lat)
    assert rel_err(host(l2[-1]), fx["reconstruction_levels"][-1]) < RECON_TOL


@pytest.mark.parametrize("name", ["residual_small", "residual_mid", "residual_full"])
def test_residual_sae(name):
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    model = load(ResidualQuantizedSAE(m["D"], m["H"], 32, abs_range=m["abs_range"], n_bits=m["n_bits"]), sd)
    assert model.sae_hidden_dims == m["hidden_dims"]
    groups, levels = model(dev(fx.x()))
    stages = [dict(enc_w=sd[f"saes.{i}.encoder.0.weight"], enc_b=sd[f"saes.{i}.encoder.0.bias"],
                   dec_w=sd[f"saes.{i}.decoder.weight"], dec_wm=sd[f"saes.{i}.decoder.weight_mirror"],
                   dec_bias=sd[f"saes.{i}.decoder.bias"]) for i in range(m["n_bits"])]
    want = oracle.residual_forward(fx.x(), stages, abs_range=m["abs_range"])
    np.testing.assert_allclose([float(g) for g in groups], want["latent_groups"], rtol=2e-3)
    # near-cutoff audit (tests/golden_util.py): a row whose stages 0..i keep every pre-activation clear of the
    # `latent > 0.5` cutoff has the same bits whatever the encoder's summation order, so its level-i output must match
    # the oracle AND the reference at 1e-5; any other row may differ by a flipped dictionary row (times 2^i)
    n_clear = 0
    for i in range(m["n_bits"]):
        clear = residual_clear_rows(want["cutoff_distance"], i)
        n_clear += int(clear.sum())
        for ref_levels in (want["reconstruction_levels"], fx["reconstruction_levels"]):
            errs = row_rel_err(host(levels[i]), ref_levels[i])
            if clear.any():
                assert errs[clear].max() < RECON_TOL, (i, errs[clear].max())
            assert errs.max() < 5e-3, i
    assert n_clear >= m["n_bits"] * m["B"] // 2                                   # the audit covers most rows


# ---- wrapper face ---------------------------------------------------------------------------
def _small_entry(name, tmp_path, model_kwargs, sd):
    entry = F.SAE_REGISTRY[name]
    path = tmp_path / f"{name}.pth"
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, path)
    return dataclasses.replace(entry, checkpoint_path=path, kwargs=model_kwargs)


def test_wrapper_binary(tmp_path, monkeypatch):
    fx = Fixture("binary_small")
    m, sd = fx.meta, fx.state_dict()
    entry = _small_entry("b_sae", tmp_path, {"input_dim": m["D"], "hidden_dim": m["H"], "gamma": m["gamma"],
                                             "n_bits": m["n_bits"]}, sd)
    monkeypatch.setitem(F.SAE_REGISTRY, "b_sae", entry)
    sae = F.load_sae("b_sae", device=DEV)
    assert isinstance(sae, F.SAEWrapper) and sae.device == torch.device(DEV)
    x = torch.from_numpy(fx.x())                      # host batch: wrapper moves it to the device
    out = sae(x)
    assert set(out) == {"latent", "reconstruction", "aux"} and set(out["aux"]) == {"polarize_loss"}
    assert out["latent"].shape == (m["B"], m["H"]) and out["reconstruction"].shape == (m["B"], m["D"])
    assert rel_err(host(out["reconstruction"]), fx["reconstruction"]) < RECON_TOL
    assert torch.equal(sae.reconstruct([x]), out["reconstruction"])          # list batch -> first element
    recs = list(sae.reconstruct_loader([x[:3], (x[3:],)]))
    assert torch.equal(torch.cat(recs), out["reconstruction"])
    det = next(iter(sae.reconstruct_loader([x], return_details=True)))
    assert set(det) == {"latent", "reconstruction", "aux"}
    dd = sae.decoder_dictionary(quantized=True)
    assert set(dd) == {"weight", "bias"} and dd["weight"].device.type == "cpu"
    step = m["gamma"] / 2 ** (m["n_bits"] - 1)
    assert np.array_equal(dd["weight"].numpy(), (step * fx["int_weights"].astype(np.float32)).astype(np.float32))
    mse = F.compute_reconstruction_error(sae, [x[:4], x[4:]])
    assert mse == pytest.approx(float(fx["mse"]), rel=1e-5)
    with pytest.raises(ValueError):
        sae([])
    with pytest.raises(TypeError):
        sae("not a tensor")


def test_wrapper_other_variants(tmp_path, monkeypatch):
    fxq = Fixture("matryoshka_small")
    mq = fxq.meta
    entry = _small_entry("q_sae", tmp_path, {"input_dim": mq["D"], "hidden_dim": mq["H"], "top_k": 32,
                                             "abs_range": mq["abs_range"], "n_bits": mq["n_bits"],
                                             "allow_bias": True}, fxq.state_dict())
    monkeypatch.setitem(F.SAE_REGISTRY, "q_sae", entry)
    sae = F.load_sae("q_sae", device=DEV)
    out = sae(torch.from_numpy(fxq.x()))
    assert set(out) == {"latent_groups", "reconstruction_levels", "reconstruction"}
    assert out["reconstruction"] is out["reconstruction_levels"][-1]
    dd = sae.decoder_dictionary()
    assert set(dd) == {"weight", "weight_mirror", "effective_weight", "bias"}
    assert torch.equal(dd["effective_weight"], dd["weight"] + dd["weight_mirror"])

    fxb = Fixture("baseline_small")
    mb = fxb.meta
    entry = _small_entry("baseline_sae", tmp_path, {"input_dim": mb["D"], "hidden_dim": mb["H"]}, fxb.state_dict())
    monkeypatch.setitem(F.SAE_REGISTRY, "baseline_sae", entry)
    sae = F.load_sae("baseline_sae", device=DEV)
    out = sae(torch.from_numpy(fxb.x()))
    assert set(out) == {"latent", "reconstruction"}
    assert rel_err(host(out["reconstruction"]), fxb["reconstruction"]) < RECON_TOL
    assert set(sae.decoder_dictionary()) == {"weight", "bias"}

    fxr = Fixture("residual_small")
    mr = fxr.meta
    entry = _small_entry("rq_sae", tmp_path, {"input_dim": mr["D"], "hidden_dim": mr["H"], "top_k": 32,
                                              "abs_range": mr["abs_range"], "n_bits": mr["n_bits"]},
                         fxr.state_dict())
    monkeypatch.setitem(F.SAE_REGISTRY, "rq_sae", entry)
    sae = F.load_sae("rq_sae", device=DEV)
    out = sae(torch.from_numpy(fxr.x()))
    assert len(out["reconstruction_levels"]) == mr["n_bits"]
    dd = sae.decoder_dictionary()
    assert "level_0_bias" in dd and "level_3_effective_weight" in dd


def test_full_size_properties():
    """Size-independent properties at the BASELINE.json shape (H=32768, a 4096-row slice of the
    65536-row batch): k non-zeros per row; kept values are exactly the encoder's; every dropped
    latent is <= the smallest kept one; linearity of the decode in the kept values."""
    from quantizedsae_amd import ops
    D, H, B = 512, 32768, 4096
    g = torch.Generator(device=DEV); g.manual_seed(5)
    model = BinarySAE(D, H, gamma=4.0, n_bits=4).to(DEV).eval()
    with torch.no_grad():
        model.decoder.weight.copy_(torch.where(torch.rand((H, D * 4), device=DEV, generator=g) > 0.5, 30.0, -30.0))
    x = torch.randn((B, D), device=DEV, generator=g)
    latent, recon, _ = model(x)
    k = model.top_k
    assert k == 65
    assert ((latent != 0).sum(1) == k).all()
    full = ops.encode_dense(x, model.encoder[0].weight, model.encoder[0].bias)
    kept = latent != 0
    assert torch.equal(latent[kept], full[kept])
    thr = torch.where(kept, latent, torch.full_like(latent, float("inf"))).min(1).values
    assert (torch.where(kept, torch.full_like(full, -float("inf")), full).max(1).values <= thr).all()
    idx, val, recon_c = model.forward_compact(x)
    assert torch.equal(recon_c, recon)
    # decode is linear in val (power-of-two scaling is exact in fp32, bias is zero here)
    r2 = model.decoder.decode_sparse(idx, val * 2)
    assert torch.equal(r2, recon * 2)
    # every latent path returns bit-identical tensors (fp16 prefilter, exact fp32 fused sweep, in-place)
    model.latent_path = "fused"
    lat_f, rec_f, _ = model(x)
    assert torch.equal(lat_f, latent) and torch.equal(rec_f, recon)
    del lat_f, rec_f
    model.latent_path = "inplace"
    lat_i, rec_i, _ = model(x)
    assert torch.equal(lat_i, latent) and torch.equal(rec_i, recon)
    del lat_i, rec_i
    model.latent_path = "auto"
    # spot-check 64 rows against the CPU oracle
    sel = torch.arange(0, B, B // 64, device=DEV)
    want = oracle.binary_forward(host(x[sel]), host(model.encoder[0].weight), host(model.encoder[0].bias),
                                 host(model.decoder.weight), host(model.decoder.bias), n_bits=4, gamma=4.0, k=k)
    assert np.array_equal(host(recon[sel]), want["reconstruction"])
    assert np.array_equal(host(latent[sel]), want["latent"])


def test_baseline_full_size_paths_agree():
    """BaselineSparseAutoencoder at H=32768 on a batch large enough for the default (fp16 candidate pass) path:
    forward(), forward_compact() and the exact-fp32 paths return the same bits; 32 rows against the oracle."""
    D, H, B = 512, 32768, 4096
    g = torch.Generator(device=DEV); g.manual_seed(6)
    model = BaselineSparseAutoencoder(D, H).to(DEV).eval()
    with torch.no_grad():
        model.encoder[0].bias.copy_(torch.randn((H,), device=DEV, generator=g) * 0.05)
    x = torch.randn((B, D), device=DEV, generator=g)
    h, recon = model(x)                                     # auto -> prefilter
    assert ((h != 0).sum(1) == model.topk).all()
    idx, val, recon_c = model.forward_compact(x)
    assert torch.equal(recon_c, recon)
    dense_c = torch.zeros_like(h).scatter_(1, idx.long(), val)
    assert torch.equal(dense_c, h)
    for path in ("fused", "inplace"):
        model.latent_path = path
        h2, r2 = model(x)
        assert torch.equal(h2, h) and torch.equal(r2, recon), path
        del h2, r2
    model.latent_path = "auto"
    sel = torch.arange(0, B, B // 32, device=DEV)
    want = oracle.baseline_forward(host(x[sel]), host(model.encoder[0].weight), host(model.encoder[0].bias),
                                   host(model.decoder.weight), host(model.decoder.bias), k=model.topk)
    assert np.array_equal(host(h[sel]), want["latent"])
    assert np.array_equal(host(recon[sel]), want["reconstruction"])


def test_wrapper_loader_pipelines_large_batches_without_changing_outputs():
    """SAEWrapper.reconstruct_loader (framework.py:325-334) over several 4096-row batches: the two-call forward runs one batch
    ahead on the loader; reconstructions, and with return_details the whole dicts, equal the batch-by-batch calls bit for bit;
    reconstruct() takes the compact path (no dense latent) with the same reconstruction bits."""
    sd = S.binary_sae_params(120, 512, 8192, 4, 30.0, 0.05, 0.1)
    model = load(BinarySAE(512, 8192, gamma=4.0, n_bits=4), sd)
    sae = F.SAEWrapper(F.SAE_REGISTRY["b_sae"], model, DEV)
    batches = [dev(S.activations(121 + i, 4096, 512)) for i in range(4)] + [dev(S.activations(130, 100, 512))]
    want = [sae(b) for b in batches]
    recs = list(sae.reconstruct_loader(batches))
    assert len(recs) == 5 and all(torch.equal(r, w["reconstruction"]) for r, w in zip(recs, want))
    dets = list(sae.reconstruct_loader(batches, return_details=True))
    for d, w in zip(dets, want):
        assert set(d) == set(w) and torch.equal(d["latent"], w["latent"]) and torch.equal(d["reconstruction"], w["reconstruction"])
    assert torch.equal(sae.reconstruct(batches[1]), want[1]["reconstruction"])
    base = F.SAEWrapper(F.SAE_REGISTRY["baseline_sae"], BaselineSparseAutoencoder(512, 8192).to(DEV), DEV)
    wb = [base(b)["reconstruction"] for b in batches]
    assert all(torch.equal(r, w) for r, w in zip(base.reconstruct_loader(batches), wb))
    mat = QuantizedMatryoshkaSAE(512, 8192, top_k=32, abs_range=4, n_bits=4).to(DEV)
    with torch.no_grad():
        mat.encoder[0].bias.fill_(-0.8)
    q = F.SAEWrapper(F.SAE_REGISTRY["q_sae"], mat, DEV)
    wq = [q(b) for b in batches]
    for d, w in zip(q.reconstruct_loader(batches, return_details=True), wq):
        assert all(torch.equal(a, b_) for a, b_ in zip(d["reconstruction_levels"], w["reconstruction_levels"]))
