"""CPU oracle for the quantized-SAE forward hot path -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of ``oracle/qsae_oracle.c`` (see that file's header for the
arithmetic contract and the reference file:line each function follows).  May be
imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg.  The product package ``quantizedsae_amd`` never imports it.

Parity status: pinned by golden vectors generated from the reference's own classes
(tools/gen_golden.py -> tests/golden/*.npz).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libqsae_oracle.so"


def build(force: bool = False) -> Path:
    """Compile the C oracle with gcc (idempotent)."""
    src = _HERE / "qsae_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libqsae_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.qsae_oracle_dot_chain.restype = C.c_float
        _lib.qsae_oracle_polarize.restype = C.c_double
        _lib.qsae_oracle_sq_err_sum.restype = C.c_double
        _lib.qsae_oracle_sigmoid_gt_cutoff.restype = C.c_float
        _lib.qsae_oracle_sigmoid_ge_cutoff.restype = C.c_float
        _lib.qsae_oracle_binary_row_bytes.restype = C.c_int
        _lib.qsae_oracle_num_threads.restype = C.c_int
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def num_threads() -> int:
    return int(lib().qsae_oracle_num_threads())


def sigmoid_cutoffs():
    """(w_gt, w_ge): sigmoid(w) > 0.5 <=> w >= w_gt ; sigmoid(w) >= 0.5 <=> w >= w_ge (fp32)."""
    return (np.float32(lib().qsae_oracle_sigmoid_gt_cutoff()),
            np.float32(lib().qsae_oracle_sigmoid_ge_cutoff()))


def encode(x, W, bias=None, act: int = ACT_NONE) -> np.ndarray:
    """latent[b,h] = act(fmaf-chain_k(x[b,k]*W[h,k]) seeded with bias[h])."""
    x, W = _f32(x), _f32(W)
    B, D = x.shape
    H, D2 = W.shape
    assert D == D2
    out = np.empty((B, H), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().qsae_oracle_encode(_p(x), _p(W), _p(b) if b is not None else None,
                             C.c_int(B), C.c_int(D), C.c_int(H), C.c_int(act), _p(out))
    return out


def dot_chain(x, w, bias: float = 0.0) -> np.float32:
    x, w = _f32(x), _f32(w)
    return np.float32(lib().qsae_oracle_dot_chain(_p(x), _p(w), C.c_float(bias), C.c_int(x.size)))


def topk(latent, k: int):
    """(idx int32 [B,k], val f32 [B,k]) ordered by (value desc, index asc)."""
    latent = _f32(latent)
    B, H = latent.shape
    assert 0 < k <= H
    idx = np.empty((B, k), dtype=np.int32)
    val = np.empty((B, k), dtype=np.float32)
    lib().qsae_oracle_topk(_p(latent), C.c_int(B), C.c_int(H), C.c_int(k), _p(idx), _p(val))
    return idx, val


def topk_gap(latent, k: int) -> np.ndarray:
    latent = _f32(latent)
    B, H = latent.shape
    gap = np.empty((B,), dtype=np.float32)
    lib().qsae_oracle_topk_gap(_p(latent), C.c_int(B), C.c_int(H), C.c_int(k), _p(gap))
    return gap


def densify(idx, val, H: int) -> np.ndarray:
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = _f32(val)
    B, k = idx.shape
    dense = np.empty((B, H), dtype=np.float32)
    lib().qsae_oracle_densify(_p(idx), _p(val), C.c_int(B), C.c_int(k), C.c_int(H), _p(dense))
    return dense


def binary_row_bytes(D: int, n_bits: int) -> int:
    return int(lib().qsae_oracle_binary_row_bytes(C.c_int(D), C.c_int(n_bits)))


def pack_binary(logits, D: int, n_bits: int) -> np.ndarray:
    """logits [H, D*n_bits] -> packed uint8 [H, row_bytes] (binary.py:49-58)."""
    logits = _f32(logits)
    H = logits.shape[0]
    assert logits.shape[1] == D * n_bits
    packed = np.empty((H, binary_row_bytes(D, n_bits)), dtype=np.uint8)
    lib().qsae_oracle_pack_binary(_p(logits), C.c_int(H), C.c_int(D), C.c_int(n_bits), _p(packed))
    return packed


def unpack_binary(packed, D: int, n_bits: int) -> np.ndarray:
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    H = packed.shape[0]
    w = np.empty((H, D), dtype=np.float32)
    lib().qsae_oracle_unpack_binary(_p(packed), C.c_int(H), C.c_int(D), C.c_int(n_bits), _p(w))
    return w


def polarize(logits, D: int, n_bits: int) -> float:
    logits = _f32(logits)
    return float(lib().qsae_oracle_polarize(_p(logits), C.c_int(logits.shape[0]), C.c_int(D), C.c_int(n_bits)))


def soft_table(logits, D: int, n_bits: int) -> np.ndarray:
    """int_weights of binary_decoder.forward (binary.py:26-35): sum_b sigmoid(logit_b) * bw_b with
    bw = [1, 2, .., -2^(n-1)], every operation rounded to fp32, bits summed LSB first."""
    logits = _f32(logits)
    H = logits.shape[0]
    assert logits.shape[1] == D * n_bits
    one = np.float32(1.0)
    with np.errstate(over="ignore"):
        p = (one / (one + np.exp(-logits, dtype=np.float32))).astype(np.float32).reshape(H, D, n_bits)
    acc = np.zeros((H, D), dtype=np.float32)
    for b in range(n_bits):
        bw = np.float32(-(1 << b) if b == n_bits - 1 else (1 << b))
        acc = (acc + (p[:, :, b] * bw).astype(np.float32)).astype(np.float32)
    return acc


def soft_gap(logits, D: int, n_bits: int) -> float:
    """max |soft integer - hard integer| over the dictionary (what qsae_pack_binary reports)."""
    hard = unpack_binary(pack_binary(logits, D, n_bits), D, n_bits)
    return float(np.max(np.abs(soft_table(logits, D, n_bits) - hard)))


def decode_binary(idx, val, packed, D: int, n_bits: int, step: float, bias=None) -> np.ndarray:
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = _f32(val)
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    B, k = idx.shape
    recon = np.empty((B, D), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().qsae_oracle_decode_binary(_p(idx), _p(val), C.c_int(B), C.c_int(k), _p(packed), C.c_int(D),
                                    C.c_int(n_bits), C.c_float(step), _p(b) if b is not None else None,
                                    _p(recon))
    return recon


def decode_table(idx, val, table, scale: float = 1.0, bias=None) -> np.ndarray:
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = _f32(val)
    table = _f32(table)
    B, k = idx.shape
    D = table.shape[1]
    recon = np.empty((B, D), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().qsae_oracle_decode_table(_p(idx), _p(val), C.c_int(B), C.c_int(k), _p(table), C.c_int(D),
                                   C.c_float(scale), _p(b) if b is not None else None, _p(recon))
    return recon


def ternary_codes(w) -> np.ndarray:
    w = _f32(w)
    codes = np.empty(w.shape, dtype=np.int8)
    lib().qsae_oracle_ternary_codes(_p(w), C.c_size_t(w.size), _p(codes))
    return codes


def decode_ternary(h, codes) -> np.ndarray:
    h = _f32(h)
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    B, H = h.shape
    D = codes.shape[0]
    assert codes.shape[1] == H
    recon = np.empty((B, D), dtype=np.float32)
    lib().qsae_oracle_decode_ternary(_p(h), C.c_int(B), C.c_int(H), _p(codes), C.c_int(D), _p(recon))
    return recon


def matryoshka_sizes(H: int, n_bits: int) -> list:
    sizes = np.zeros((n_bits,), dtype=np.int32)
    lib().qsae_oracle_matryoshka_sizes(C.c_int(H), C.c_int(n_bits), _p(sizes))
    return [int(s) for s in sizes]


def matryoshka_pack(w, wm, n_bits: int, abs_range: float):
    w, wm = _f32(w), _f32(wm)
    H, D = w.shape
    codes = np.empty((H, D), dtype=np.int8)
    scale = np.empty((H,), dtype=np.float32)
    lib().qsae_oracle_matryoshka_pack(_p(w), _p(wm), C.c_int(H), C.c_int(D), C.c_int(n_bits),
                                      C.c_float(abs_range), _p(codes), _p(scale))
    return codes, scale


def zbits(pre) -> np.ndarray:
    pre = _f32(pre)
    bits = np.empty(pre.shape, dtype=np.uint8)
    lib().qsae_oracle_zbits(_p(pre), C.c_size_t(pre.size), _p(bits))
    return bits


def decode_matryoshka(zb, codes, scale, bias, n_bits: int, allow_bias: bool = True):
    """-> (levels f32 [n,B,D] cumulative, l0 f32 [n])."""
    zb = np.ascontiguousarray(zb, dtype=np.uint8)
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    scale = _f32(scale)
    B, H = zb.shape
    D = codes.shape[1]
    levels = np.empty((n_bits, B, D), dtype=np.float32)
    l0 = np.empty((n_bits,), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().qsae_oracle_decode_matryoshka(_p(zb), C.c_int(B), C.c_int(H), C.c_int(D), C.c_int(n_bits),
                                        _p(codes), _p(scale), _p(b) if b is not None else None,
                                        C.c_int(1 if allow_bias else 0), _p(levels), _p(l0))
    return levels, l0


def sq_err_sum(recon, x) -> float:
    recon, x = _f32(recon), _f32(x)
    assert recon.shape == x.shape
    return float(lib().qsae_oracle_sq_err_sum(_p(recon), _p(x), C.c_size_t(recon.size)))


# ---------------------------------------------------------------------------
# Whole-forward restatements (compose the primitives exactly as the reference does)

def binary_forward(x, enc_w, enc_b, dec_logits, dec_bias, *, n_bits: int, gamma: float, k: int = None,
                   soft: bool = False):
    """BinarySAE.forward (binary.py:91-103).  soft=False: the hard two's-complement integers of
    quantized_int_weights() (binary.py:49-58) -- what the forward converges to on a polarised checkpoint;
    soft=True: the sigmoid-bit integers the reference forward multiplies with (binary.py:26-38), evaluated on the k
    kept entries in ascending index order.
    Returns dict(idx, val, latent(dense), reconstruction, polarize_loss)."""
    H, D = np.asarray(enc_w).shape
    if k is None:
        k = int(H * 0.002)
    latent = encode(x, enc_w, enc_b, ACT_NONE)
    idx, val = topk(latent, k)
    step = np.float32(gamma / (2 ** (n_bits - 1)))
    if soft:
        recon = decode_table(idx, val, soft_table(dec_logits, D, n_bits), float(step), dec_bias)
    else:
        packed = pack_binary(dec_logits, D, n_bits)
        recon = decode_binary(idx, val, packed, D, n_bits, float(step), dec_bias)
    return {"idx": idx, "val": val, "latent_full": latent, "latent": densify(idx, val, H),
            "reconstruction": recon, "polarize_loss": polarize(dec_logits, D, n_bits)}


def baseline_forward(x, enc_w, enc_b, dec_w, dec_b, *, k: int = 32):
    """BaselineSparseAutoencoder.forward (baseline.py:17-40); dec_w is [D,H]."""
    H, D = np.asarray(enc_w).shape
    latent = encode(x, enc_w, enc_b, ACT_NONE)
    idx, val = topk(latent, k)
    table = np.ascontiguousarray(np.asarray(dec_w, dtype=np.float32).T)
    recon = decode_table(idx, val, table, 1.0, dec_b)
    return {"idx": idx, "val": val, "latent_full": latent, "latent": densify(idx, val, H),
            "reconstruction": recon}


def ternary_forward(x, enc_w, enc_b, dec_w):
    """TernarySparseAutoencoder.forward (ternary.py:116-122, 41-52); dec_w is [D,H]."""
    h = encode(x, enc_w, enc_b, ACT_RELU)
    codes = ternary_codes(dec_w)
    return {"latent": h, "reconstruction": decode_ternary(h, codes)}


def matryoshka_forward(x, enc_w, enc_b, dec_w, dec_wm, dec_bias, *, n_bits: int, abs_range: float,
                       allow_bias: bool = True):
    """QuantizedMatryoshkaSAE.forward (quantized_matryoshka.py:217-220, 47-143)."""
    pre = encode(x, enc_w, enc_b, ACT_NONE)
    zb = zbits(pre)
    codes, scale = matryoshka_pack(dec_w, dec_wm, n_bits, abs_range)
    levels, l0 = decode_matryoshka(zb, codes, scale, dec_bias, n_bits, allow_bias)
    gt, _ge = sigmoid_cutoffs()
    # per row: how close the nearest pre-activation comes to the `latent > 0.5` cutoff (a bit that another summation
    # order of the encoder could flip); used by the tests' near-cutoff audit
    dist = np.min(np.abs(pre.astype(np.float64) - np.float64(gt)), axis=1)
    return {"zbits": zb, "latent_groups": l0, "reconstruction_levels": levels,
            "reconstruction": levels[-1], "cutoff_distance": dist}


def residual_forward(x, stages, *, abs_range: float):
    """ResidualQuantizedSAE.forward (residual_quantized.py:53-69).  `stages` is a list of
    dicts(enc_w, enc_b, dec_w, dec_wm, dec_bias); each stage is a 1-bit matryoshka SAE,
    bias only in stage 0; residual = (residual - recon) * 2."""
    residual = np.ascontiguousarray(x, dtype=np.float32)
    groups, levels, dist = [], [], []
    for i, st in enumerate(stages):
        out = matryoshka_forward(residual, st["enc_w"], st["enc_b"], st["dec_w"], st["dec_wm"],
                                 st["dec_bias"], n_bits=1, abs_range=abs_range, allow_bias=(i == 0))
        groups.append(out["latent_groups"][-1])
        levels.append(out["reconstruction"])
        dist.append(out["cutoff_distance"])
        residual = ((residual - out["reconstruction"]).astype(np.float32) * np.float32(2)).astype(np.float32)
    # cutoff_distance[i][b]: nearest pre-activation of stage i, row b, to the cutoff.  A row whose stages 0..i all stay
    # clear of it has the same bits under any summation order of the encoder, hence the same level-i output up to
    # rounding; a row that does not may legitimately differ by one dictionary row times 2^i (residuals are doubled).
    return {"latent_groups": np.asarray(groups, dtype=np.float32),
            "reconstruction_levels": np.stack(levels), "reconstruction": levels[-1],
            "cutoff_distance": np.stack(dist)}


# ---- consumers of the sparse latent (scripts/analysis/dynamic_analysis.py:255-311) -------------------------
def activation_stats(mask: np.ndarray):
    """(activation_counts int64 [H], coactivation int32 [H, H]) of a boolean activation mask [B, H]:
    ``mask.sum(dim=0)`` and ``mask_int.t() @ mask_int`` (dynamic_analysis.py:292-296)."""
    m = np.asarray(mask).astype(np.int32)
    return m.sum(axis=0).astype(np.int64), (m.T @ m).astype(np.int32)


# ---- activation quantizer of the binary datasets (src/quantized_sae/data/dataset.py:76-102) ------------------
def quantize_bits(x: np.ndarray, n_bits: int, scale_factor: float, signed: bool = True) -> np.ndarray:
    """LSB-first 0/1 floats [B, D * n_bits] of the n-bit activation codes; fp32 arithmetic step by step,
    round half to even, NaN -> INT_MIN like torch's .int()."""
    x = np.asarray(x, dtype=np.float32)
    sf = np.float32(scale_factor)
    t = (x * sf).astype(np.float32)
    half = np.float32(2 ** (n_bits - 1))
    if signed:
        lo, hi = -half, half - np.float32(1)
    else:
        t = (t * np.float32(2)).astype(np.float32)
        t = (t + half).astype(np.float32)
        lo, hi = np.float32(0), np.float32(2 ** n_bits - 1)
    with np.errstate(invalid="ignore"):
        c = np.where(t < lo, lo, t)
        c = np.where(c > hi, hi, c)
        q = np.where(np.isnan(c), np.int64(-2 ** 31), np.rint(c).astype(np.int64))
    u = (q & (2 ** n_bits - 1)).astype(np.uint32)
    bits = ((u[..., None] >> np.arange(n_bits, dtype=np.uint32)) & 1).astype(np.float32)
    return bits.reshape(x.shape[0], -1)


# ---- BinaryLatentSAE (sae/binary_latent.py:6-28) -----------------------------------------------------------
def binary_latent_forward(x, enc_w, enc_b, dec_w, dec_b) -> dict:
    """binary_latent = (sigmoid(encoder pre-activation) >= 0.5) as 0/1 floats; reconstruction = decoder Linear of
    the binary latent (the reference feeds latent + (binary - latent), equal to it up to one rounding)."""
    pre = encode(x, enc_w, enc_b)
    _gt, ge = sigmoid_cutoffs()
    binary = (pre >= ge).astype(np.float32)
    return {"pre": pre, "binary_latent": binary, "reconstruction": encode(binary, dec_w, dec_b)}
