"""StyleGAN (v1) generator and its Noise layer on the HIP kernels.

Reference: modelling/stylegan.py -- same class names, constructor signatures, child-module layout and state_dict keys
(`Noise.scale`; `StyleGanGeneratorBlock` = ModuleList [AdaIN, LeakyReLU, Upsample, Conv2d, Noise, LeakyReLU, Conv2d,
Noise]; `StyleGanGenerator.{mapping_network, learned_input, blocks, out_convs}`).  Fusions per block: AdaIN + LeakyReLU
is one affine pass behind a statistics kernel; Upsample -> Conv2d -> Noise -> LeakyReLU and Conv2d -> Noise are one
implicit-GEMM launch each (nearest 2x in the gather, bias / noise / activation on the accumulators); the RGB skip
`interpolate(out) + out_conv(x)` (stylegan.py:84) is one pass.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .. import functional as GF
from .common import AdaIN, Conv2d, LeakyReLU, Upsample, to_act


class Noise(nn.Module):
    """x + randn(N, 1, H, W) * scale[c] (stylegan.py:19-27): one draw per forward from the global generator, same
    shape / dtype / order as the reference; the add runs in the fused affine kernel."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(dim, 1, 1))

    def forward(self, x: Tensor) -> Tensor:
        x = to_act(x)
        n, _, h, w = x.shape
        noise = torch.randn(n, 1, h, w, device=x.device, dtype=x.dtype)
        return GF.affine_act(x, noise=noise, ns=self.scale)


def _plain_conv_fused(conv: nn.Conv2d, x: Tensor, noise_mod: "Noise", slope: float, up2: bool) -> Tensor:
    """[Upsample ->] Conv2d -> Noise [-> LeakyReLU] as one launch (the modulated-convolution kernel without styles)."""
    from .stylegan2 import ModConvFn

    n, _, h, w = x.shape
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    draw = torch.randn(n, 1, ho, wo, device=x.device, dtype=x.dtype)  # the Noise layer's draw (stylegan.py:26)
    return ModConvFn.apply(x, conv.weight, None, None, conv.bias, draw, noise_mod.scale, slope, up2)


class StyleGanGeneratorBlock(nn.ModuleList):
    def __init__(self, in_dim: int, out_dim: int, z_dim: int) -> None:
        layers = [
            AdaIN(in_dim, z_dim),
            LeakyReLU(0.2),
            Upsample(scale_factor=2.0),
            Conv2d(in_dim, out_dim, 3, 1, 1),
            Noise(out_dim),
            LeakyReLU(0.2),
            Conv2d(out_dim, out_dim, 3, 1, 1),
            Noise(out_dim),
        ]
        super().__init__(layers)

    def forward(self, x: Tensor, w: Tensor) -> Tensor:
        adain, act0, up, conv1, noise1, act1, conv2, noise2 = list(self)
        x = adain.fused(to_act(x), w, act0.negative_slope)
        x = _plain_conv_fused(conv1, x, noise1, act1.negative_slope, True)   # leaky-ReLU and nearest 2x commute
        return _plain_conv_fused(conv2, x, noise2, 1.0, False)


class StyleGanGenerator(nn.Module):
    def __init__(self, img_size: int, img_channels: int = 3, z_dim: int = 128, base_dim: int = 16, mapping_network_depth: int = 8) -> None:
        super().__init__()
        self.mapping_network = nn.Sequential(nn.LayerNorm(z_dim))
        for _ in range(mapping_network_depth):
            self.mapping_network.extend([nn.Linear(z_dim, z_dim), nn.LeakyReLU(0.2)])

        self.learned_input = nn.Parameter(torch.ones(1, 512, 4, 4))
        self.blocks = nn.ModuleList()
        self.out_convs = nn.ModuleList()
        in_ch = 512

        depth = int(math.log2(img_size // 4))
        for i in range(depth):
            out_ch = min(base_dim * img_size // 4 // 2 ** (i + 1), 512)
            self.blocks.append(StyleGanGeneratorBlock(in_ch, out_ch, z_dim))
            in_ch = out_ch
            self.out_convs.append(Conv2d(in_ch, img_channels, 1))

        self.reset_parameters()

    def reset_parameters(self):
        from .progressive_gan import init_weights

        self.apply(init_weights)

    def forward(self, z: Tensor) -> Tensor:
        with torch.autocast("cuda", enabled=False):  # (N, 128) x (128, 128) products: plain library GEMMs in fp32
            w = GF.mapping_network(self.mapping_network, z.float()) * 0.01
        dt = GF.act_dtype(z)
        x = self.learned_input.to(dt).expand(z.shape[0], -1, -1, -1)
        x = self.blocks[0](x, w)
        out = self.out_convs[0](x)
        for block, out_conv in zip(self.blocks[1:], self.out_convs[1:]):
            x = block(x, w)
            out = GF.up2_add(out, out_conv(x))  # F.interpolate(out, scale_factor=2.0) + out_conv(x), one pass
        return out
