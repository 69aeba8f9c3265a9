"""Fused decoder pieces on the gfx950 kernels: norm, cache, attention, MLP, MoE, block, model."""
