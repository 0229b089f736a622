"""upfirdn2d / Blur / fused bias + leaky-ReLU on channels-last HBM-streaming HIP kernels.

Reference: modelling/nvidia_ops.py.  Same names, positional signatures and error behaviour:
    upfirdn2d(imgs, kernel, up, down, px1, px2, py1, py2)      :18-34
    UpFIRDn2d.apply(imgs, kernel, up, down, px1, px2, py1, py2) :40-54
    Blur(kernel_size=3, up=1, down=1), Blur.make_kernel         :58-78
    up_conv_blur / blur_conv_down                               :81-123
`bias_leaky_relu` is additive API (the reference has no fused function): it equals
F.leaky_relu(x + b.view(1,-1,1,1), 0.2) -- no sqrt(2) gain, no clamp.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn

from .. import functional as GF
from .common import AvgPool2d, Conv2d, Upsample, to_act

# The reference's autograd.Function: `UpFIRDn2d.apply(...)` with the same 8 positionals.  Every
# order of gradient re-enters the same HIP kernel with flipped taps and up/down swapped.
UpFIRDn2d = GF.UpFirDnFn


def upfirdn2d(imgs: Tensor, kernel: Tensor, up: int, down: int, px1: int, px2: int, py1: int, py2: int) -> Tensor:
    """Zero-insert x`up`, pad (negative pads crop), correlate with the unflipped `kernel`, stride `down`."""
    return GF.upfirdn2d(to_act(imgs), kernel, up, down, px1, px2, py1, py2)


def bias_leaky_relu(x: Tensor, bias: Optional[Tensor] = None, negative_slope: float = 0.2) -> Tensor:
    return GF.bias_leaky_relu(to_act(x), bias, negative_slope)


class Blur(nn.Module):
    def __init__(self, kernel_size: int = 3, up: int = 1, down: int = 1):
        super().__init__()
        self.up = up
        self.down = down
        self.p1 = (kernel_size - 1) // 2
        self.p2 = kernel_size - 1 - self.p1
        self.register_buffer("kernel", self.make_kernel(kernel_size))
        self.kernel: Tensor

    def forward(self, imgs: Tensor) -> Tensor:
        return upfirdn2d(imgs, self.kernel, self.up, self.down, self.p1, self.p2, self.p1, self.p2)

    @staticmethod
    def make_kernel(kernel_size: int) -> Tensor:
        assert 2 <= kernel_size <= 7
        row = torch.tensor([math.comb(kernel_size - 1, i) for i in range(kernel_size)], dtype=torch.float)
        row = row / row.sum()
        return row.view(1, -1) * row.view(-1, 1)


class _ZeroStuffConvTranspose2d(nn.ConvTranspose2d):
    """ConvTranspose2d(stride 2) = zero-insert x2 (upfirdn2d with a unit tap) + stride-1 correlation
    with the transposed, flipped weights."""

    def forward(self, x: Tensor) -> Tensor:
        k = self.kernel_size[0]
        p = self.padding[0]
        op = self.output_padding[0]
        x = to_act(x)
        one = torch.ones(1, 1, device=x.device)
        full = k - 1 - p
        x = GF.upfirdn2d(x, one, 2, 1, full, full - 1 + op, full, full - 1 + op)
        w = self.weight.flip(2, 3).transpose(0, 1)
        y = GF.Conv2dFn.apply(x, w, 0, 0)
        return GF.affine_act(y, b=self.bias) if self.bias is not None else y


class _SubsampledConv2d(nn.Conv2d):
    """Conv2d(stride 2) = stride-1 correlation + keep every second sample."""

    def forward(self, x: Tensor) -> Tensor:
        x = to_act(x)
        y = GF.Conv2dFn.apply(x, self.weight, self.padding[0], self.padding[1])
        one = torch.ones(1, 1, device=x.device)
        y = GF.upfirdn2d(y, one, 1, 2, 0, 0, 0, 0)
        return GF.affine_act(y, b=self.bias) if self.bias is not None else y


def up_conv_blur(in_channels: int, out_channels: int, kernel_size: int, blur_size: Optional[int] = 3):
    padding = (kernel_size - 1) // 2
    output_padding = kernel_size % 2
    if blur_size is None:
        return nn.Sequential(Upsample(scale_factor=2.0), Conv2d(in_channels, out_channels, kernel_size, 1, padding))
    if kernel_size == 1:
        return nn.Sequential(Conv2d(in_channels, out_channels, 1), Blur(blur_size, up=2))
    return nn.Sequential(
        _ZeroStuffConvTranspose2d(in_channels, out_channels, kernel_size, 2, padding, output_padding), Blur(blur_size)
    )


def blur_conv_down(in_channels: int, out_channels: int, kernel_size: int, blur_size: Optional[int] = 3):
    padding = (kernel_size - 1) // 2
    if blur_size is None:
        return nn.Sequential(Conv2d(in_channels, out_channels, kernel_size, 1, padding), AvgPool2d(2))
    if kernel_size == 1:
        return nn.Sequential(Blur(blur_size, down=2), Conv2d(in_channels, out_channels, 1))
    return nn.Sequential(Blur(blur_size), _SubsampledConv2d(in_channels, out_channels, kernel_size, 2, padding))
