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
import torch.nn.functional as F

from .common import AvgPool2d, Conv2d, Upsample, _ConvTranspose2d, _StridedConv2d, to_act

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


class _ZeroStuffConvTranspose2d(_ConvTranspose2d):
    """ConvTranspose2d(k, stride 2, (k-1)//2, output_padding k % 2) of up_conv_blur (nvidia_ops.py:93-101).  Odd k: the
    kernel gets a zero row and column (k -> k + 1, output_padding 1 -> 0: the same map, the same output size) and runs as
    the adjoint of a space-to-depth folded convolution (common._ConvTranspose2d) -- no zero-stuffed 4x tensor, 1.8x the
    useful products instead of 4x.  The name is round 1's (zero insertion + stride-1 correlation)."""

    def forward(self, x: Tensor) -> Tensor:
        k, op = self.kernel_size[0], self.output_padding[0]
        if self.stride == (2, 2) and k % 2 == 1 and op == 1 and k == self.kernel_size[1]:
            return self._forward_with(x, F.pad(self.weight, (0, 1, 0, 1)), k + 1, 0)
        return _ConvTranspose2d.forward(self, x)


class _SubsampledConv2d(_StridedConv2d):
    """Conv2d(k, stride 2, (k-1)//2) of blur_conv_down (nvidia_ops.py:116-122).  Odd k: a zero row and column make the
    kernel even, and the layer runs as space-to-depth + a stride-1 valid convolution (common._StridedConv2d) instead of a
    stride-1 convolution whose result is subsampled (4x the products).  The name is round 1's."""

    def forward(self, x: Tensor) -> Tensor:
        k = self.kernel_size[0]
        if self.stride == (2, 2) and k % 2 == 1 and k == self.kernel_size[1] and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            return self._forward_with(x, F.pad(self.weight, (0, 1, 0, 1)), k + 1)
        return _StridedConv2d.forward(self, x)


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
