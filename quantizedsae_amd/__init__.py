"""quantizedsae_amd -- MI355X (gfx950) inference backend for quantized sparse autoencoders.

Drop-in for the forward path of ASSERT-KTH/QuantizedSAE: the module classes in
``quantizedsae_amd.sae`` keep the reference constructors, ``forward()`` signatures and
state_dict keys; ``quantizedsae_amd.inference`` keeps ``load_sae`` / ``SAEWrapper`` /
``SAE_REGISTRY``.  All compute runs in hand-written HIP kernels (``csrc/``) reached through
the C ABI of ``include/qsae.h``; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"

from .sae import (  # noqa: F401
    BaselineSparseAutoencoder,
    BinarySAE,
    QuantizedMatryoshkaSAE,
    ResidualQuantizedSAE,
    SparseAutoencoder,
    TernarySparseAutoencoder,
)
from .inference import SAE_REGISTRY, SAEWrapper, available_saes, load_sae  # noqa: F401
