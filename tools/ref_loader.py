"""Load the ASSERT-KTH/QuantizedSAE hot-path classes from /root/reference by file path.

Container-only tooling (the reference never travels to the GPU box).  The package
does not import as shipped (`sae/binary.py:7-8` imports the pre-refactor module
names `baseSAE.SAE` / `nnba.adder`; `inference/framework.py:9-12` imports `SAEs.*`),
so the files are loaded one by one under those legacy names.
Nothing from the reference is copied into this repo; this only *executes* it here
to produce golden vectors (tools/gen_golden.py).
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

REF_ROOT = Path("/root/reference")
_SAE = REF_ROOT / "src" / "quantized_sae" / "sae"
_INF = REF_ROOT / "src" / "quantized_sae" / "inference"


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference classes (BinarySAE, ...)."""
    if not REF_ROOT.exists():
        raise FileNotFoundError(f"{REF_ROOT} is not present (reference is container-only)")
    for pkg in ("baseSAE", "nnba", "SAEs"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # mark as package
            sys.modules[pkg] = m
    base = _load("baseSAE.SAE", _SAE / "base.py")
    sys.modules["nnba.adder"] = types.ModuleType("nnba.adder")  # `import *` of nothing
    baseline = _load("SAEs.baseline_SAE", _SAE / "baseline.py")
    binary = _load("SAEs.binary_SAE", _SAE / "binary.py")
    qm = _load("SAEs.quantized_matryoshka_SAE", _SAE / "quantized_matryoshka.py")
    rq = _load("SAEs.residual_quantized_matryoshka_SAE", _SAE / "residual_quantized.py")
    ternary = _load("ref_ternary", _SAE / "ternary.py")
    blatent = _load("ref_binary_latent", _SAE / "binary_latent.py")
    dataset = _load("ref_dataset", REF_ROOT / "src" / "quantized_sae" / "data" / "dataset.py")
    framework = _load("ref_framework", _INF / "framework.py")
    ns = types.SimpleNamespace(
        SparseAutoencoder=base.SparseAutoencoder,
        BaselineSparseAutoencoder=baseline.BaselineSparseAutoencoder,
        BinarySAE=binary.BinarySAE,
        binary_decoder=binary.binary_decoder,
        QuantizedMatryoshkaSAE=qm.QuantizedMatryoshkaSAE,
        QuantizedMatryoshkaDecoder=qm.QuantizedMatryoshkaDecoder,
        ResidualQuantizedSAE=rq.ResidualQuantizedSAE,
        TernarySparseAutoencoder=ternary.TernarySparseAutoencoder,
        STEWeights=ternary.STEWeights,
        BinaryLatentSAE=blatent.BinaryLatentSAE,
        dataset=dataset,
        framework=framework,
    )
    return ns


if __name__ == "__main__":
    ns = load_reference()
    print("reference classes loaded:", [k for k in vars(ns)])
