"""Noise module of the StyleGAN family (reference: modelling/stylegan.py:19-27)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import functional as GF
from .common import to_act


class Noise(nn.Module):
    """x + randn(N,1,H,W) * scale[C,1,1]; scale zero-initialised.  Draws exactly one
    torch.randn(N, 1, H, W, device=x.device, dtype=x.dtype) from the global RNG per forward, like
    the reference, so seeded runs consume the generator in the same order."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(dim, 1, 1))

    def draw(self, x: Tensor) -> Tensor:
        n, _, h, w = x.shape
        return torch.randn(n, 1, h, w, device=x.device, dtype=x.dtype)

    def forward(self, x: Tensor) -> Tensor:
        x = to_act(x)
        return GF.affine_act(x, noise=self.draw(x), ns=self.scale)
