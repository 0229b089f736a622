"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Deterministic pseudo-random tensors built from integer arithmetic only, so the golden
generator (this container) and the tests (GPU box) reconstruct bit-identical large
parameter sets without committing tens of MB of weights or relying on a torch RNG stream.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping

import numpy as np
import torch


def det_uniform(shape, salt: int) -> torch.Tensor:
    """float32 tensor in [-0.5, 0.5): v[i] = hash32(i, salt) / 2^32 - 0.5 (exact in fp64 -> fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    x = (idx + np.uint64(salt & 0xFFFFFFFF) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    # xorshift-multiply mixer on 32-bit lanes held in uint64 (all ops exact)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    v = x.astype(np.float64) / 4294967296.0 - 0.5
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def _salt(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x85EBCA6B)) & 0xFFFFFFFF


def det_state(shapes: Mapping[str, tuple], seed: int, *, bias_amp: float = 0.1, scale_amp: float = 0.1) -> Dict[str, torch.Tensor]:
    """Parameters for a shapes dict: weights uniform with the kaiming variance 2/fan_in, biases and
    Noise/D `scale` tensors uniform with the given amplitudes (non-zero on purpose, so every term of
    every gradient is exercised), `learned_input` 1 + 0.5*u, LayerNorm weight 1 + 0.2*u."""
    out = {}
    for name, shape in shapes.items():
        u = det_uniform(shape, _salt(name, seed))
        if name == "learned_input":
            t = 1 + 0.5 * u
        elif name.startswith("mapping_network.0.") and name.endswith("weight"):
            t = 1 + 0.2 * u
        elif name.endswith("weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = u * math.sqrt(12.0) * math.sqrt(2.0 / fan_in)
        elif name.endswith("weight") and len(shape) == 1:  # LayerNorm / LayerNorm2d gains
            t = 1 + 0.2 * u
        elif name.endswith("scale"):
            t = u * 2 * scale_amp
        else:
            t = u * 2 * bias_amp
        out[name] = t.contiguous()
    return out


def recipe_state(shapes: Mapping[str, tuple], seed: int, scale_amp: float, wgain: float) -> Dict[str, torch.Tensor]:
    """det_state for the trainer-recipe fixtures (tests/golden/trainer_recipes.pt): biases +-0.05, matrices / kernels x wgain."""
    sd = det_state({k: tuple(v) for k, v in shapes.items()}, seed=seed, bias_amp=0.05, scale_amp=scale_amp)
    return {k: (v * wgain if v.ndim >= 2 else v) for k, v in sd.items()}


def projections(t: torch.Tensor, salt: int = 7) -> torch.Tensor:
    """Compact fingerprint of a big tensor: [sum, abs-sum, <t, r1>, <t, r2>] in float64."""
    t64 = t.detach().double().flatten().cpu()
    r1 = det_uniform((t64.numel(),), salt).double()
    r2 = det_uniform((t64.numel(),), salt + 1).double()
    return torch.stack([t64.sum(), t64.abs().sum(), (t64 * r1).sum(), (t64 * r2).sum()])
