#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes (container-only).

The reference (ASSERT-KTH/QuantizedSAE) ships no tests, fixtures or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself:
this script loads the reference classes from /root/reference (tools/ref_loader.py),
feeds them parameters/inputs from the portable PRNG in quantizedsae_amd/synthetic.py and
stores inputs' *recipe* (seed + shapes; the arrays themselves when small) together with
the reference outputs.  Only data is written -- no reference source text.

Run:  python tools/gen_golden.py            (needs /root/reference; CPU only, ~2 min)
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from quantizedsae_amd import synthetic as S  # noqa: E402
from ref_loader import load_reference  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_sd(model, sd, strict=True):
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=strict)
    model.eval()
    return model


def save(name, meta, **arrays):
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"  wrote {path.name}: {path.stat().st_size/1024:.1f} KiB  keys={list(arrays)}")


def topk_gap(latent: np.ndarray, k: int) -> np.ndarray:
    s = -np.sort(-latent, axis=1)
    return (s[:, k - 1] - s[:, k]).astype(np.float32) if k < latent.shape[1] else np.full(latent.shape[0], np.inf, np.float32)


# ---------------------------------------------------------------------------
def gen_binary(ref, name, seed, D, H, B, n_bits, gamma, k=None, store_inputs=False, rows=None,
               enc_bias_std=0.0, dec_bias_std=0.0, logit_std=None):
    """logit_std: unpolarised decoder logits (the reference forward then differs from a hard-bit decode: it IS the soft
    sigmoid-bit computation, binary.py:24-47); the fixture also stores the soft integer table."""
    sd = S.binary_sae_params(seed, D, H, n_bits, enc_bias_std=enc_bias_std, dec_bias_std=dec_bias_std,
                             logit_std=logit_std)
    x = S.activations(seed, B, D)
    model = ref.BinarySAE(D, H, gamma=gamma, n_bits=n_bits)
    load_sd(model, sd)
    if k is not None:
        model.k = k / H          # forward uses int(hidden_dim * self.k)   (binary.py:94)
    kk = int(H * model.k)
    with torch.no_grad():
        sparse_latent, recon, pol = model(t(x))
        latent_full = model.encode(t(x)).numpy()
        int_w = model.decoder.quantized_int_weights().numpy()
        soft_w = model.decoder.quantized_int_weights_continuous().numpy()
    sparse_latent = sparse_latent.numpy()
    # index set of the kept entries, straight from torch.topk inside the reference
    with torch.no_grad():
        vals, idxs = model.encode(t(x)).topk(kk, dim=1)
    mse = float(((recon - t(x)) ** 2).sum().item() / recon.numel())
    r = slice(None) if rows is None else slice(0, rows)
    meta = dict(variant="binary", seed=seed, D=D, H=H, B=B, n_bits=n_bits, gamma=gamma, k=kk,
                rows=(B if rows is None else rows), logit_mag=30.0, enc_bias_std=enc_bias_std,
                dec_bias_std=dec_bias_std, torch=torch.__version__)
    if logit_std is not None:
        meta["logit_std"] = logit_std
    arrays = dict(topk_idx=np.sort(idxs.numpy()[r], axis=1).astype(np.int32),
                  topk_val_sorted_by_idx=np.take_along_axis(vals.numpy()[r], np.argsort(idxs.numpy()[r], axis=1), axis=1),
                  gap=topk_gap(latent_full, kk)[r],
                  reconstruction=recon.numpy()[r], polarize_loss=np.float64(pol.item()),
                  mse=np.float64(mse))
    if logit_std is not None:
        arrays.update(soft_gap=np.float64(np.abs(soft_w - int_w).max()))
        if H * D <= 1 << 17:
            arrays.update(soft_int_weights=soft_w.astype(np.float32))
    if store_inputs:
        arrays.update(x=x, **{"sd." + k_: v for k_, v in sd.items() if k_ != "decoder.weight"},
                      dec_bits=np.packbits((int_w.astype(np.int64)[..., None] >> np.arange(n_bits) & 1).astype(np.uint8)
                                           .reshape(H, D * n_bits), axis=1),
                      int_weights=int_w.astype(np.int8), sparse_latent=sparse_latent)
    save(name, meta, **arrays)


def gen_binary_soft(ref):
    """Unpolarised BinarySAE checkpoints: N(0, 2^2) logits, the reference's default (kaiming) decoder init, and a
    mid-size model at logit std 1 -- outputs of the reference forward, which uses the soft sigmoid bits."""
    gen_binary(ref, "binary_soft_small", seed=16, D=64, H=1000, B=7, n_bits=4, gamma=4.0, store_inputs=True,
               enc_bias_std=0.05, dec_bias_std=0.2, logit_std=2.0)
    gen_binary(ref, "binary_soft_init", seed=17, D=64, H=1000, B=7, n_bits=4, gamma=4.0, store_inputs=True,
               logit_std=float(np.sqrt(2.0 / (64 * 4))))
    gen_binary(ref, "binary_soft_n8", seed=19, D=32, H=512, B=5, n_bits=8, gamma=1.5, k=3, store_inputs=True,
               dec_bias_std=0.2, logit_std=3.0)
    gen_binary(ref, "binary_soft_mid", seed=18, D=512, H=2048, B=32, n_bits=4, gamma=4.0, k=65,
               enc_bias_std=0.05, dec_bias_std=0.2, logit_std=1.0)


def gen_baseline(ref, name, seed, D, H, B, store_inputs=False, rows=None, bias_std=0.0):
    sd = S.baseline_sae_params(seed, D, H, bias_std=bias_std)
    x = S.activations(seed, B, D)
    model = ref.BaselineSparseAutoencoder(D, H)
    load_sd(model, sd)
    with torch.no_grad():
        h, recon = model(t(x))
        latent_full = model.encoder(t(x))
        vals, idxs = torch.topk(latent_full, model.topk, dim=1)
    r = slice(None) if rows is None else slice(0, rows)
    mse = float(((recon - t(x)) ** 2).sum().item() / recon.numel())
    meta = dict(variant="baseline", seed=seed, D=D, H=H, B=B, k=model.topk, bias_std=bias_std,
                rows=(B if rows is None else rows), torch=torch.__version__)
    arrays = dict(topk_idx=np.sort(idxs.numpy()[r], axis=1).astype(np.int32),
                  topk_val_sorted_by_idx=np.take_along_axis(vals.numpy()[r], np.argsort(idxs.numpy()[r], axis=1), axis=1),
                  gap=topk_gap(latent_full.numpy(), model.topk)[r],
                  reconstruction=recon.numpy()[r], mse=np.float64(mse))
    if store_inputs:
        arrays.update(x=x, sparse_latent=h.numpy(), **{"sd." + k_: v for k_, v in sd.items()})
    save(name, meta, **arrays)


def gen_ternary(ref, name, seed, D, H, B, store_inputs=False, latent_rows=None):
    """latent_rows: store only the first rows of the dense [B, H] latent (full-size fixture) plus per-row digests."""
    sd = S.ternary_sae_params(seed, D, H)
    x = S.activations(seed, B, D)
    model = ref.TernarySparseAutoencoder(D, H)
    load_sd(model, sd, strict=False)   # input_activations/output_grad buffers are None at init
    with torch.no_grad():
        h, recon = model(t(x))
    meta = dict(variant="ternary", seed=seed, D=D, H=H, B=B, w_std=0.5, torch=torch.__version__)
    hn = h.numpy()
    arrays = dict(latent=hn if latent_rows is None else hn[:latent_rows], reconstruction=recon.numpy(),
                  latent_nnz=(hn > 0).sum(1).astype(np.int32), latent_sum=hn.astype(np.float64).sum(1),
                  nonzero_code_fraction=np.float64((np.abs(sd["decoder.weight"]) >= 0.5).mean()))
    if store_inputs:
        arrays.update(x=x, **{"sd." + k_: v for k_, v in sd.items() if k_ != "decoder.mask"})
    save(name, meta, **arrays)


def gen_matryoshka(ref, name, seed, D, H, B, n_bits, abs_range, store_inputs=False, min_abs=1e-3,
                   enc_bias_shift=None, bias_std=0.3, edge=False):
    sd = S.matryoshka_sae_params(seed, D, H, enc_bias_shift=enc_bias_shift, min_abs=min_abs, bias_std=bias_std)
    if edge:
        # logits straddling the fp32 sigmoid cutoffs: |w| < 1e-7 including the exact cutoff values
        tiny = np.array([0.0, -0.0, 5e-8, -5e-8, 8.9406967e-08, 8.9406974e-08, -1.788139e-07,
                         -1.7881392e-07, 1e-7, -1e-7, 1.2e-7, -2e-7], dtype=np.float32)
        pick = (S.hash_u64(seed, H * D, stream=77) % np.uint64(len(tiny))).astype(np.int64).reshape(H, D)
        pick2 = (S.hash_u64(seed, H * D, stream=78) % np.uint64(len(tiny))).astype(np.int64).reshape(H, D)
        sd["decoder.weight"] = tiny[pick]
        sd["decoder.weight_mirror"] = tiny[pick2]
    x = S.activations(seed, B, D)
    model = ref.QuantizedMatryoshkaSAE(D, H, top_k=32, abs_range=abs_range, n_bits=n_bits)
    load_sd(model, sd)
    with torch.no_grad():
        groups, levels = model(t(x))
        lat = model.encoder(t(x))
        zb = (lat > 0.5).numpy().astype(np.uint8)
    meta = dict(variant="matryoshka", seed=seed, D=D, H=H, B=B, n_bits=n_bits, abs_range=abs_range,
                min_abs=min_abs, enc_bias_shift=enc_bias_shift, bias_std=bias_std, edge=edge,
                sizes=[int(s) for s in model.decoder.nested_dictionary_size], torch=torch.__version__)
    arrays = dict(latent_groups=np.array([g.item() for g in groups], dtype=np.float32),
                  reconstruction_levels=np.stack([l.numpy() for l in levels]),
                  zbits=np.packbits(zb, axis=1))
    if store_inputs:
        arrays.update(x=x, **{"sd." + k_: v for k_, v in sd.items()})
    save(name, meta, **arrays)


def gen_residual(ref, name, seed, D, H, B, n_bits, abs_range, store_inputs=False):
    model = ref.ResidualQuantizedSAE(D, H, top_k=32, abs_range=abs_range, n_bits=n_bits)
    sd = {}
    for i, hdim in enumerate(model.sae_hidden_dims):
        sub = S.matryoshka_sae_params(seed, D, hdim, enc_bias_sigmas=-1.5, bias_std=(0.3 if i == 0 else 0.0),
                                      stream0=100 * (i + 1))
        for k_, v in sub.items():
            sd[f"saes.{i}.{k_}"] = v
    load_sd(model, sd)
    x = S.activations(seed, B, D)
    with torch.no_grad():
        groups, levels = model(t(x))
    meta = dict(variant="residual", seed=seed, D=D, H=H, B=B, n_bits=n_bits, abs_range=abs_range,
                hidden_dims=[int(s) for s in model.sae_hidden_dims], torch=torch.__version__)
    arrays = dict(latent_groups=np.array([g.item() for g in groups], dtype=np.float32),
                  reconstruction_levels=np.stack([l.numpy() for l in levels]))
    if store_inputs:
        arrays.update(x=x, **{"sd." + k_: v for k_, v in sd.items()})
    save(name, meta, **arrays)


def gen_sigmoid_cutoffs():
    """Re-measure the fp32 cutoffs of torch.sigmoid by bisection over float bit patterns."""
    def toi(f):
        u = int(np.float32(f).view(np.uint32))
        return u if u < 0x80000000 else -(u & 0x7FFFFFFF)

    def tof(i):
        u = i if i >= 0 else ((-i) | 0x80000000)
        return np.uint32(u).view(np.float32)

    def first_true(pred, lo, hi):
        a, b = toi(lo), toi(hi)
        while b - a > 1:
            m = (a + b) // 2
            if pred(tof(m)):
                b = m
            else:
                a = m
        return tof(b)
    gt = first_true(lambda w: bool(torch.sigmoid(torch.full((64,), float(w)))[0] > 0.5), 0.0, 1e-5)
    ge = first_true(lambda w: bool(torch.sigmoid(torch.full((64,), float(w)))[0] >= 0.5), -1e-5, 1e-5)
    save("sigmoid_cutoffs", dict(torch=torch.__version__),
         gt_cutoff_bits=np.uint32(gt.view(np.uint32)), ge_cutoff_bits=np.uint32(ge.view(np.uint32)))


def main():
    ref = load_reference()
    only = set(sys.argv[1:])
    if only == {"binary_soft"}:
        return gen_binary_soft(ref)
    if only == {"full"}:
        gen_ternary(ref, "ternary_full", seed=2, D=512, H=32768, B=32, latent_rows=2)
        return gen_residual(ref, "residual_full", seed=53, D=512, H=32768, B=16, n_bits=4, abs_range=1.5)
    if only and not (only & {"binary", "baseline", "ternary"}):
        return main_tail(ref, only)
    print("sigmoid cutoffs")
    gen_sigmoid_cutoffs()
    print("binary")
    gen_binary(ref, "binary_small", seed=11, D=64, H=1000, B=7, n_bits=4, gamma=4.0, store_inputs=True,
               enc_bias_std=0.05, dec_bias_std=0.2)
    gen_binary(ref, "binary_n8", seed=12, D=32, H=512, B=5, n_bits=8, gamma=1.5, k=3, store_inputs=True,
               dec_bias_std=0.2)
    gen_binary(ref, "binary_n2", seed=15, D=48, H=600, B=4, n_bits=2, gamma=4.0, k=5, store_inputs=True)
    gen_binary(ref, "binary_mid", seed=13, D=512, H=2048, B=32, n_bits=4, gamma=4.0, k=65,
               enc_bias_std=0.05, dec_bias_std=0.2)
    gen_binary(ref, "binary_full_g4", seed=1, D=512, H=32768, B=256, n_bits=4, gamma=4.0)
    gen_binary(ref, "binary_full_g15", seed=1, D=512, H=32768, B=64, n_bits=4, gamma=1.5)
    gen_binary_soft(ref)
    print("baseline")
    gen_baseline(ref, "baseline_small", seed=21, D=64, H=1000, B=7, store_inputs=True, bias_std=0.1)
    gen_baseline(ref, "baseline_mid", seed=22, D=512, H=2048, B=32, bias_std=0.1)
    gen_baseline(ref, "baseline_full", seed=0, D=512, H=32768, B=256)
    print("ternary")
    gen_ternary(ref, "ternary_small", seed=31, D=64, H=1000, B=7, store_inputs=True)
    gen_ternary(ref, "ternary_mid", seed=32, D=512, H=2048, B=16)
    gen_ternary(ref, "ternary_full", seed=2, D=512, H=32768, B=32, latent_rows=2)
    main_tail(ref, only)


def gen_binary_latent(ref, name, seed, D, H, B):
    """BinaryLatentSAE (sae/binary_latent.py:6-28): sigmoid encoder, latent >= 0.5, dense decoder."""
    sd = {"encoder.0.weight": S.xavier_uniform(seed, H, D, stream=1),
          "encoder.0.bias": S.normal(seed, (H,), stream=2, std=0.05),
          "decoder.weight": S.uniform(seed, (D, H), -1.0 / np.sqrt(H), 1.0 / np.sqrt(H), stream=3),
          "decoder.bias": S.normal(seed, (D,), stream=4, std=0.1)}
    x = S.activations(seed, B, D)
    model = ref.BinaryLatentSAE(D, H)
    load_sd(model, sd)
    with torch.no_grad():
        binary_latent, recon = model(t(x))
        pre = torch.nn.functional.linear(t(x), t(sd["encoder.0.weight"]), t(sd["encoder.0.bias"]))
    save(name, dict(variant="binary_latent", seed=seed, D=D, H=H, B=B),
         x=x, binary_latent=np.packbits(binary_latent.numpy().astype(np.uint8), axis=1),
         reconstruction=recon.numpy(), pre_min_abs_margin=np.abs(pre.numpy() + 1.7881390590446244e-07).min(axis=1),
         **{f"sd.{k}": v for k, v in sd.items()})


def gen_quantize_bits(ref):
    """n-bit activation quantizers of the binary dataset class (data/dataset.py:76-102), called unbound on a stub
    that carries the attributes the methods read."""
    import types
    cls = ref.dataset.HiddenStatesTorchDatasetInBinary
    x = (S.normal(71, (48, 64), stream=1, std=2.0)).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1e-9, 3.99999, 4.0, -4.0, 100.0, -100.0]
    out = {}
    for n_bits, gamma in ((4, 4), (8, 1.5), (2, 4)):
        stub = types.SimpleNamespace(n_bits=n_bits, gamma=gamma, shift_factor=2 ** (n_bits - 1),
                                     scale_factor=2 ** (n_bits - 1) / (gamma + 1e-5))
        for nm in ("quantize", "quantize_signed"):
            rows = [getattr(cls, nm)(stub, t(r)).numpy() for r in x]
            out[f"{nm}_n{n_bits}"] = np.packbits(np.stack(rows).astype(np.uint8), axis=1)
    save("quantize_bits", dict(variant="quantize_bits", configs=[[4, 4], [8, 1.5], [2, 4]]), x=x, **out)


def main_tail(ref, only):
    if not only or "extras" in only:
        print("binary latent / quantizers")
        gen_binary_latent(ref, "binary_latent_small", seed=61, D=64, H=1000, B=9)
        gen_quantize_bits(ref)
        if only == {"extras"}:
            return
    print("matryoshka")
    gen_matryoshka(ref, "matryoshka_small", seed=41, D=64, H=1000, B=7, n_bits=4, abs_range=4.0, store_inputs=True)
    gen_matryoshka(ref, "matryoshka_edge", seed=42, D=64, H=512, B=5, n_bits=4, abs_range=1.5, store_inputs=True,
                   edge=True)
    gen_matryoshka(ref, "matryoshka_mid", seed=43, D=512, H=2048, B=16, n_bits=4, abs_range=4.0)
    gen_matryoshka(ref, "matryoshka_full", seed=3, D=512, H=32768, B=32, n_bits=4, abs_range=4.0)
    print("residual")
    gen_residual(ref, "residual_small", seed=51, D=64, H=1024, B=7, n_bits=4, abs_range=1.5, store_inputs=True)
    gen_residual(ref, "residual_mid", seed=52, D=512, H=4096, B=8, n_bits=4, abs_range=1.5)
    gen_residual(ref, "residual_full", seed=53, D=512, H=32768, B=16, n_bits=4, abs_range=1.5)


if __name__ == "__main__":
    main()
