"""Unified inference interface (reference: src/quantized_sae/inference/)."""
from .framework import (  # noqa: F401
    SAE_REGISTRY,
    SAERegistryEntry,
    SAEWrapper,
    available_saes,
    compute_reconstruction_error,
    load_sae,
)
