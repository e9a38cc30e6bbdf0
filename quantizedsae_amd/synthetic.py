"""Portable synthetic workload generator.

A counter-based integer hash (splitmix64 finaliser) drives every synthetic tensor so
that the build container, the GPU box and the golden-vector generator produce
bit-identical inputs without relying on any library RNG stream or libm: only exact
integer ops, int->float conversion and IEEE add/mul are used.

"normal" is a 12-uniform Irwin-Hall sum (unit variance, support +-6) -- a synthetic
stand-in for N(0,1) activations; it is *not* torch.randn.
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, n: int, stream: int = 0, offset: int = 0) -> np.ndarray:
    """n 64-bit hashes of counters offset..offset+n-1 under (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * _GOLD + np.uint64(stream) * _M2 + np.uint64(0x1234567))
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        return _mix(ctr * _GOLD + base)


def uniform01(seed: int, n: int, stream: int = 0, offset: int = 0) -> np.ndarray:
    """float64 uniforms in [0,1) with 53 random bits (exact arithmetic)."""
    return (hash_u64(seed, n, stream, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed: int, shape, lo: float, hi: float, stream: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(seed, n, stream)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, shape, stream: int = 0, std: float = 1.0, chunk: int = 1 << 22) -> np.ndarray:
    """Irwin-Hall(12) - 6: unit-variance bell curve from 12 exact uniforms per sample."""
    n = int(np.prod(shape))
    out = np.empty((n,), dtype=np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        acc = np.zeros((m,), dtype=np.float64)
        for j in range(12):
            acc += uniform01(seed, m, stream * 16 + j + 1, offset=s)
        out[s:s + m] = ((acc - 6.0) * std).astype(np.float32)
    return out.reshape(shape)


def fair_bits(seed: int, shape, stream: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return ((hash_u64(seed, n, stream) >> np.uint64(63)).astype(np.uint8)).reshape(shape)


# ---------------------------------------------------------------------------
# Model parameter sets of SURVEY.md section 8(d) (numpy, state_dict key names of the reference)

def xavier_uniform(seed: int, fan_out: int, fan_in: int, stream: int = 0) -> np.ndarray:
    bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return uniform(seed, (fan_out, fan_in), -bound, bound, stream)


def binary_sae_params(seed: int, D: int, H: int, n_bits: int, logit_mag: float = 30.0,
                      enc_bias_std: float = 0.0, dec_bias_std: float = 0.0, logit_std: float = None) -> dict:
    """BinarySAE state_dict (binary.py:73-89): xavier encoder, saturated +-logit_mag decoder bits -- or, with
    logit_std, unpolarised bell-shaped logits of that standard deviation (logit_std = sqrt(2 / (D n_bits)) is the
    reference's default kaiming init, binary.py:22)."""
    if logit_std is not None:
        dec_w = normal(seed, (H, D * n_bits), stream=2, std=logit_std)
    else:
        bits = fair_bits(seed, (H, D * n_bits), stream=2).astype(np.float32)
        dec_w = ((bits * 2.0 - 1.0) * np.float32(logit_mag)).astype(np.float32)
    sd = {
        "encoder.0.weight": xavier_uniform(seed, H, D, stream=1),
        "encoder.0.bias": (normal(seed, (H,), stream=3, std=enc_bias_std) if enc_bias_std else np.zeros((H,), np.float32)),
        "decoder.weight": dec_w,
        "decoder.bias": (normal(seed, (D,), stream=4, std=dec_bias_std) if dec_bias_std else np.zeros((D,), np.float32)),
    }
    return sd


def baseline_sae_params(seed: int, D: int, H: int, bias_std: float = 0.0) -> dict:
    """BaselineSparseAutoencoder state_dict (baseline.py:4-15)."""
    b = 1.0 / float(np.sqrt(H))
    return {
        "encoder.0.weight": xavier_uniform(seed, H, D, stream=1),
        "encoder.0.bias": (normal(seed, (H,), stream=3, std=bias_std) if bias_std else np.zeros((H,), np.float32)),
        "decoder.weight": uniform(seed, (D, H), -b, b, stream=2),
        "decoder.bias": (normal(seed, (D,), stream=4, std=bias_std) if bias_std else np.zeros((D,), np.float32)),
    }


def ternary_sae_params(seed: int, D: int, H: int, w_std: float = 0.5) -> dict:
    """TernarySparseAutoencoder state_dict (ternary.py:92-100); decoder w ~ bell(0, w_std)
    so that ~32% of the codes are non-zero (default kaiming init gives all zeros)."""
    b = 1.0 / float(np.sqrt(D))
    return {
        "encoder.0.weight": uniform(seed, (H, D), -b, b, stream=1),
        "encoder.0.bias": uniform(seed, (H,), -b, b, stream=3),
        "decoder.weight": normal(seed, (D, H), stream=2, std=w_std),
        "decoder.mask": np.ones((D, H), np.float32),
    }


def matryoshka_sae_params(seed: int, D: int, H: int, enc_bias_shift: float = None,
                          min_abs: float = 1e-3, bias_std: float = 0.0, stream0: int = 0,
                          enc_bias_sigmas: float = -2.5) -> dict:
    """QuantizedMatryoshkaSAE state_dict (quantized_matryoshka.py:40-45,206-212):
    decoder logits U(+-1) pushed away from 0 by min_abs; encoder bias shifted by
    enc_bias_sigmas standard deviations of the pre-activation (unit-variance inputs) so
    that L0 is sparse (~0.6% of H) instead of 50%."""
    if enc_bias_shift is None:
        enc_bias_shift = enc_bias_sigmas * float(np.sqrt(D) * np.sqrt(6.0 / (D + H)) / np.sqrt(3.0))
    def away(a):
        return np.where(np.abs(a) < min_abs, np.where(a < 0, -min_abs, min_abs), a).astype(np.float32)
    return {
        "encoder.0.weight": xavier_uniform(seed, H, D, stream=stream0 + 1),
        "encoder.0.bias": np.full((H,), enc_bias_shift, np.float32),
        "decoder.weight": away(uniform(seed, (H, D), -1.0, 1.0, stream=stream0 + 2)),
        "decoder.weight_mirror": away(uniform(seed, (H, D), -1.0, 1.0, stream=stream0 + 5)),
        "decoder.bias": (normal(seed, (D,), stream=stream0 + 4, std=bias_std) if bias_std else np.zeros((D,), np.float32)),
    }


def activations(seed: int, B: int, D: int, stream: int = 9) -> np.ndarray:
    return normal(seed, (B, D), stream=stream)


# ---------------------------------------------------------------------------
# Config 5 stand-ins (BASELINE.json: "pythia-70m-deduped activation stream"; the real dumps do not exist offline)

HEAVY_DIMS = (7, 100, 300, 511)        # residual-stream "massive activation" dimensions of the heavy-tailed stand-in


def heavy_tailed_activations(seed: int, B: int, D: int, stream: int = 9, dim_scale: float = 30.0,
                             row_spread: float = 10.0) -> np.ndarray:
    """Bell-shaped rows with (i) four dimensions at dim_scale times the scale of the rest -- the few outlier
    dimensions transformer residual streams carry -- and (ii) a per-row scale spread over a factor row_spread
    (log-uniform): rows of very different norm in one batch.  What this stresses: the prefilter's error budget
    eps_b grows with ||x_b|| max||W_h||, so outlier dimensions and heavy encoder rows widen every unit's margin."""
    x = normal(seed, (B, D), stream=stream)
    for d in HEAVY_DIMS:
        if d < D:
            x[:, d] *= np.float32(dim_scale)
    u = uniform01(seed, B, stream=stream * 16 + 15)
    x *= np.exp((u - 0.5) * np.log(row_spread)).astype(np.float32)[:, None]
    return x


def heavy_tailed_binary_sae_params(seed: int, D: int, H: int, n_bits: int, logit_mag: float = 30.0) -> dict:
    """binary_sae_params with encoder biases != 0 (bell, std 0.1) and 1 % of the encoder rows at 5 times their norm."""
    sd = binary_sae_params(seed, D, H, n_bits, logit_mag, enc_bias_std=0.1, dec_bias_std=0.05)
    heavy = hash_u64(seed, H, stream=77) % np.uint64(100) == 0
    sd["encoder.0.weight"][heavy] *= np.float32(5.0)
    return sd
