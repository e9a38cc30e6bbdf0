"""SAE model classes (reference: src/quantized_sae/sae/)."""
from .base import SparseAutoencoder
from .baseline import BaselineSparseAutoencoder
from .binary import BinarySAE, binary_decoder
from .binary_latent import BinaryLatentSAE
from .quantized_matryoshka import QuantizedMatryoshkaDecoder, QuantizedMatryoshkaSAE
from .residual_quantized import ResidualQuantizedSAE
from .ternary import STEWeights, TernarySparseAutoencoder

__all__ = [
    "SparseAutoencoder", "BaselineSparseAutoencoder", "BinarySAE", "binary_decoder", "BinaryLatentSAE",
    "QuantizedMatryoshkaDecoder", "QuantizedMatryoshkaSAE", "ResidualQuantizedSAE",
    "STEWeights", "TernarySparseAutoencoder",
]
