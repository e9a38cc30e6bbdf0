"""Activation statistics from the compact / bit-packed latent: integer work, bit-exact
against the oracle's mask arithmetic and against masks the reference itself produced (golden fixtures)."""
import dataclasses

import numpy as np
import pytest
import torch

import oracle
from golden_util import Fixture
from quantizedsae_amd import BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, synthetic as S
from quantizedsae_amd.inference import analysis as A
from quantizedsae_amd.inference import framework as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def _ops():
    from quantizedsae_amd import ops
    return ops


@pytest.mark.parametrize("B,k,H", [(37, 5, 64), (300, 65, 1024), (1, 1, 32), (129, 256, 4096)])
def test_counts_and_coactivation_from_compact_rows(B, k, H):
    ops = _ops()
    rng = np.random.default_rng(B * 7 + k)
    idx = np.stack([rng.permutation(H)[:k] for _ in range(B)]).astype(np.int32)
    val = rng.standard_normal((B, k)).astype(np.float32)          # about half the entries are inactive (<= 0)
    val[0, 0] = 0.0
    mask = np.zeros((B, H), bool)
    np.put_along_axis(mask, idx.astype(np.int64), val > 0, axis=1)
    want_counts, want_co = oracle.activation_stats(mask)
    counts = ops.activation_counts(dev(idx), dev(val), H)
    coact = ops.coactivation_sparse(dev(idx), dev(val), H)
    assert np.array_equal(host(counts), want_counts)
    assert np.array_equal(host(coact), want_co)
    # accumulation over calls, and val=None counts every listed entry
    ops.activation_counts(dev(idx), dev(val), H, counts)
    assert np.array_equal(host(counts), 2 * want_counts)
    all_on = np.zeros((B, H), bool)
    np.put_along_axis(all_on, idx.astype(np.int64), True, axis=1)
    assert np.array_equal(host(ops.activation_counts(dev(idx), None, H)), all_on.sum(0))
    assert np.array_equal(host(ops.coactivation_sparse(dev(idx), None, H)), oracle.activation_stats(all_on)[1])


@pytest.mark.parametrize("B,words", [(5, 1), (333, 7), (1030, 64)])
def test_counts_from_packed_bits(B, words):
    ops = _ops()
    bits = S.fair_bits(41, (B, 32 * words))
    packed = np.packbits(bits, axis=1, bitorder="little").view(np.int32)
    got = host(ops.activation_counts_bits(dev(packed)))
    assert np.array_equal(got, bits.sum(0).astype(np.int64))


def _wrap(name, model):
    return F.SAEWrapper(F.SAE_REGISTRY[name], model, DEV)


def _load(model, sd):
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return model.to(DEV).eval()


@pytest.mark.parametrize("name", ["binary_small", "baseline_small"])
def test_activation_stats_match_reference_masks(name):
    """mask / counts / co-activation / tokens per feature against the reference's own sparse latent."""
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    if m["variant"] == "binary":
        model = _load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
        model.k = m["k"] / m["H"]
        sae = _wrap("b_sae", model)
    else:
        model = _load(BaselineSparseAutoencoder(m["D"], m["H"]), sd)
        sae = _wrap("baseline_sae", model)
    x = fx.x()[: fx["sparse_latent"].shape[0]]
    ref_mask = fx["sparse_latent"] > 0
    rows_ok = np.ones(len(x), bool)
    if "gap" in fx:                                        # rows whose k/(k+1) gap is a near-tie may pick another unit
        from golden_util import NEAR_TIE_EPS
        rows_ok = fx["gap"][: len(x)] > NEAR_TIE_EPS
    mask = A._activation_mask(sae, dev(x)).numpy()
    assert mask.dtype == np.bool_ and np.array_equal(mask[rows_ok], ref_mask[rows_ok])
    want_counts, want_co = oracle.activation_stats(mask)
    n = len(x)
    tpc = 2 if n % 2 == 0 else 1
    tokens = torch.arange(n, dtype=torch.long).reshape(-1, tpc) * 3 + 1
    cut = max(1, n // 3)
    loader = [torch.from_numpy(x[:cut]), [torch.from_numpy(x[cut:])]]        # tensors and (tensor,) batches
    st = A.compute_activation_stats(sae, loader, token_ids=tokens, tokens_per_context=tpc)
    assert np.array_equal(st["activation_counts"].numpy(), want_counts)
    assert np.array_equal(st["coactivation"].numpy(), want_co)
    flat_tok = tokens.reshape(-1).numpy()
    for f in range(m["H"]):
        assert st["tokens_per_feature"][f] == flat_tok[np.nonzero(mask[:, f])[0]].tolist()
    l0 = A.compute_l0_by_level(sae, loader)
    assert l0.shape == (1,) and float(l0[0]) == pytest.approx(mask.sum() / len(x), rel=1e-12)


def test_matryoshka_mask_and_l0_match_reference_bits():
    fx = Fixture("matryoshka_small")
    m, sd = fx.meta, fx.state_dict()
    model = _load(QuantizedMatryoshkaSAE(m["D"], m["H"], 32, abs_range=m["abs_range"], n_bits=m["n_bits"]), sd)
    sae = _wrap("q_sae", model)
    x = fx.x()
    ref_mask = np.unpackbits(fx["zbits"], axis=1)[:, : m["H"]].astype(bool)
    mask = A._activation_mask(sae, dev(x)).numpy()
    assert np.array_equal(mask, ref_mask)
    l0 = A.compute_l0_by_level(sae, [torch.from_numpy(x)]).numpy()
    bounds = np.cumsum([0] + list(m["sizes"]))
    want = np.array([ref_mask[:, bounds[i]:bounds[i + 1]].sum() / len(x) for i in range(len(m["sizes"]))])
    np.testing.assert_allclose(l0, want, rtol=1e-12)
    st = A.compute_activation_stats(sae, [torch.from_numpy(x)], token_ids=torch.zeros((len(x), 1), dtype=torch.long),
                                    tokens_per_context=1, with_tokens=False)
    want_counts, want_co = oracle.activation_stats(ref_mask)
    assert np.array_equal(st["activation_counts"].numpy(), want_counts)
    assert np.array_equal(st["coactivation"].numpy(), want_co)


@pytest.mark.parametrize("name", ["binary_small", "baseline_small", "matryoshka_small"])
def test_analyze_dataset_one_pass(name):
    """analyze_dataset (dynamic_analysis.py:317-440): one pass = the separate helpers' results + the final MSE, which
    is checked against the reference's own reconstruction (golden fixture)."""
    fx = Fixture(name)
    m, sd = fx.meta, fx.state_dict()
    if m["variant"] == "binary":
        model = _load(BinarySAE(m["D"], m["H"], gamma=m["gamma"], n_bits=m["n_bits"]), sd)
        model.k = m["k"] / m["H"]
        sae = _wrap("b_sae", model)
    elif m["variant"] == "baseline":
        model = _load(BaselineSparseAutoencoder(m["D"], m["H"]), sd)
        sae = _wrap("baseline_sae", model)
    else:
        model = _load(QuantizedMatryoshkaSAE(m["D"], m["H"], 32, abs_range=m["abs_range"], n_bits=m["n_bits"]), sd)
        sae = _wrap("q_sae", model)
    x = fx.x()
    n = len(x)
    tokens = (torch.arange(n, dtype=torch.long) * 7 + 2).reshape(n, 1)
    cut = max(1, n // 3)
    loader = [torch.from_numpy(x[:cut]), [torch.from_numpy(x[cut:])]]
    st = A.analyze_dataset(sae, loader, token_ids=tokens, tokens_per_context=1, device=DEV)
    assert set(st) == {"mse_final", "mse_per_level", "l0_per_level", "activation_counts", "coactivation", "tokens_per_feature"}
    assert st["mse_per_level"] is None and st["l0_per_level"] is None
    sep = A.compute_activation_stats(sae, loader, token_ids=tokens, tokens_per_context=1)
    assert torch.equal(st["activation_counts"], sep["activation_counts"])
    assert torch.equal(st["coactivation"], sep["coactivation"])
    assert st["tokens_per_feature"] == sep["tokens_per_feature"]
    assert st["mse_final"] == pytest.approx(A.compute_reconstruction_error(sae, loader), rel=1e-12)
    ref_recon = fx["reconstruction_levels"][-1] if m["variant"] == "matryoshka" else fx["reconstruction"]
    if len(ref_recon) == n:
        want = float(((ref_recon.astype(np.float64) - x) ** 2).mean())
        assert st["mse_final"] == pytest.approx(want, rel=2e-5)


def test_reconstruction_error_by_level():
    """matryoshka: every cumulative level against the input; residual: every stage against its own residual
    (dynamic_analysis.py:103-165); top-k models: a length-1 tensor."""
    from quantizedsae_amd import ResidualQuantizedSAE
    fx = Fixture("matryoshka_small")
    m, sd = fx.meta, fx.state_dict()
    model = _load(QuantizedMatryoshkaSAE(m["D"], m["H"], 32, abs_range=m["abs_range"], n_bits=m["n_bits"]), sd)
    sae = _wrap("q_sae", model)
    x = fx.x()
    loader = [torch.from_numpy(x[:5]), torch.from_numpy(x[5:])]
    got = A.compute_reconstruction_error_by_level(sae, loader).numpy()
    want = np.array([((lv.astype(np.float64) - x) ** 2).mean() for lv in fx["reconstruction_levels"]])
    np.testing.assert_allclose(got, want, rtol=2e-5)
    # residual SAE: against a torch restatement of the reference loop on the model's own outputs
    torch.manual_seed(3)
    rq = ResidualQuantizedSAE(64, 512, top_k=8, abs_range=1.5, n_bits=3).to(DEV).eval()
    xr = torch.randn(40, 64)
    got = A.compute_reconstruction_error_by_level(_wrap("rq_sae", rq), [xr[:16], xr[16:]])
    _, levels = rq(xr.to(DEV))
    residual, want = xr.to(DEV), []
    for recon in levels:
        want.append(float(((recon - residual).double() ** 2).mean()))
        residual = (residual - recon) * 2
    np.testing.assert_allclose(got.numpy(), np.array(want), rtol=1e-6)
    # top-k model
    fb = Fixture("baseline_small")
    base = _load(BaselineSparseAutoencoder(fb.meta["D"], fb.meta["H"]), fb.state_dict())
    one = A.compute_reconstruction_error_by_level(_wrap("baseline_sae", base), [torch.from_numpy(fb.x())])
    assert one.shape == (1,) and one.dtype == torch.float64


# ---- chunk datasets, activation quantizers, BinaryLatentSAE --------------------------------
def test_quantize_bits_kernel_matches_oracle_and_reference():
    ops = _ops()
    fx = Fixture("quantize_bits")
    x = fx["x"].copy()
    for n_bits, gamma in fx.meta["configs"]:
        sf = 2 ** (n_bits - 1) / (gamma + 1e-5)
        for nm, signed in (("quantize", False), ("quantize_signed", True)):
            want = np.unpackbits(fx[f"{nm}_n{n_bits}"], axis=1)[:, : x.shape[1] * n_bits].astype(np.float32)
            got = host(ops.quantize_bits(dev(x), n_bits, sf, signed=signed))
            assert np.array_equal(got, want), (nm, n_bits)
    big = S.normal(72, (1000, 512), stream=1, std=3.0).astype(np.float32)
    big[5, 7] = np.nan
    big[6, 8] = np.inf
    for signed in (False, True):
        assert np.array_equal(host(ops.quantize_bits(dev(big), 4, 2 ** 3 / (4 + 1e-5), signed=signed)),
                              oracle.quantize_bits(big, 4, 2 ** 3 / (4 + 1e-5), signed=signed))


def test_chunk_datasets(tmp_path):
    from quantizedsae_amd import data as Dd
    chunk = torch.from_numpy(S.normal(73, (6, 5, 64), stream=1, std=2.0).astype(np.float16))   # [ctx, tok, D] as stored
    path = tmp_path / "chunk.pt"
    torch.save(chunk, path)
    ds = Dd.HiddenStatesTorchDataset(path)
    assert len(ds) == 30 and ds.files_info[1:] == (6, 5, 64)
    flat = chunk.reshape(30, 64).float()
    assert torch.equal(ds[7], flat[7]) and ds[7].dtype == torch.float32
    # row shards of two ranks tile the chunk; batches arrive on the device as fp32
    got = [torch.cat(list(ds.iter_batches(4, DEV, world_size=2, rank=r))) for r in (0, 1)]
    assert torch.equal(torch.cat(got).cpu(), flat)
    dsb = Dd.HiddenStatesTorchDatasetInBinary(path, gamma=4, n_bits=4)
    want = oracle.quantize_bits(flat.numpy(), 4, dsb.scale_factor, signed=True)
    assert np.array_equal(dsb[3].numpy(), want[3])                           # per-sample face: quantize_signed
    assert np.array_equal(dsb.quantize(flat[3]).numpy(), oracle.quantize_bits(flat[3:4].numpy(), 4, dsb.scale_factor,
                                                                               signed=False)[0])
    assert np.array_equal(host(dsb.quantize_batch(flat.to(DEV))), want)      # batch face: one kernel
    assert torch.equal(dsb.__getoriginalitem__(3), flat[3])


def test_binary_latent_sae():
    from quantizedsae_amd.sae import BinaryLatentSAE
    fx = Fixture("binary_latent_small")
    m = fx.meta
    sd = {k[3:]: a for k, a in fx.arrays.items() if k.startswith("sd.")}
    model = _load(BinaryLatentSAE(m["D"], m["H"]), sd)
    binary, recon = model(dev(fx["x"]))
    want = oracle.binary_latent_forward(fx["x"], sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                        sd["decoder.bias"])
    assert np.array_equal(host(binary), want["binary_latent"])              # bit-exact vs the oracle
    assert np.array_equal(host(recon), want["reconstruction"])
    from golden_util import NEAR_TIE_EPS, row_rel_err
    rows_ok = fx["pre_min_abs_margin"] > NEAR_TIE_EPS
    ref_bits = np.unpackbits(fx["binary_latent"], axis=1)[:, : m["H"]].astype(np.float32)
    assert np.array_equal(host(binary)[rows_ok], ref_bits[rows_ok])         # vs the reference's own output
    assert row_rel_err(host(recon), fx["reconstruction"])[rows_ok].max() < 1e-5
