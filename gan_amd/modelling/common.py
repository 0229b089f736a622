"""Leaf modules with the reference's names, constructor signatures and state_dict keys, computing on
the HIP kernels.  Reference: modelling/common.py:27-70 plus the stock torch.nn layers the reference's
StyleGAN2 path instantiates (nn.Conv2d, nn.LeakyReLU, nn.AvgPool2d, nn.Upsample)."""
from __future__ import annotations

from functools import partial

import torch
from torch import Tensor, nn

from .. import functional as GF


def to_act(x: Tensor) -> Tensor:
    """Cast to the kernels' compute dtype (autocast dtype if autocast is on) and channels-last."""
    if x.dim() != 4:
        raise GF._lib.GanAmdError(f"expected a 4-D (N, C, H, W) tensor, got shape {tuple(x.shape)}")
    dt = GF.act_dtype(x)
    if dt not in (torch.float32, torch.bfloat16):
        raise GF._lib.GanAmdError(f"unsupported compute dtype {dt}: the kernels take float32 or bfloat16")
    if x.dtype != dt:  # cast and layout change in one copy
        return x.to(dtype=dt, memory_format=torch.channels_last)
    return x.contiguous(memory_format=torch.channels_last)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (stride 1, dilation 1, groups 1, zero padding) on the MFMA implicit-GEMM kernel."""

    def _check(self):
        if self.stride != (1, 1) or self.dilation != (1, 1) or self.groups != 1 or self.padding_mode != "zeros":
            raise GF._lib.GanAmdError("gan_amd Conv2d supports stride 1, dilation 1, groups 1, zero padding")
        if isinstance(self.padding, str):
            raise GF._lib.GanAmdError("gan_amd Conv2d takes integer padding")

    def forward(self, x: Tensor) -> Tensor:
        self._check()
        if self.padding == (0, 0) and x.is_cuda and GF.rgb_conv_shape(x, self.weight):
            # the image boundary (progressive_gan.py:61): reads the 3-channel tensor as it is -- fp32 or bf16, NCHW or
            # channels-last -- instead of a cast pass, a 16-channel padded copy and a convolution over the copy
            dt = GF.act_dtype(x)
            if dt in (torch.float32, torch.bfloat16):
                return GF.RgbConvFn.apply(x, self.weight, self.bias, 1.0, dt)
        x = to_act(x)
        if self.bias is not None and self.padding[0] == self.padding[1]:
            return GF.conv_bias_act(x, self.weight, self.bias, self.padding[0])  # bias on the accumulators
        y = GF.Conv2dFn.apply(x, self.weight, self.padding[0], self.padding[1])
        if self.bias is not None:
            y = GF.affine_act(y, b=self.bias)
        return y


class LeakyReLU(nn.LeakyReLU):
    """Always out of place: the reference's inplace=True breaks autograd in the D block;
    forward values are identical."""

    def forward(self, x: Tensor) -> Tensor:
        return GF.leaky_relu(to_act(x), self.negative_slope)


class AvgPool2d(nn.AvgPool2d):
    def forward(self, x: Tensor) -> Tensor:
        if self.kernel_size not in (2, (2, 2)):
            raise GF._lib.GanAmdError("gan_amd AvgPool2d supports kernel_size=2")
        return GF.avg_pool2(to_act(x))


class Upsample(nn.Upsample):
    def forward(self, x: Tensor) -> Tensor:
        if self.scale_factor not in (2, 2.0) or self.mode != "nearest":
            raise GF._lib.GanAmdError("gan_amd Upsample supports nearest x2")
        return GF.nearest_up2(to_act(x))


class AdaIN(nn.Module):
    """modelling/common.py:7-16: F.instance_norm(x) * (style_w(w) + 1) + style_b(w), as statistics + one affine pass."""

    def __init__(self, dim: int, z_dim: int) -> None:
        super().__init__()
        self.style_w = nn.Linear(z_dim, dim)
        self.style_b = nn.Linear(z_dim, dim)

    def fused(self, x: Tensor, w: Tensor, slope: float) -> Tensor:
        """AdaIN followed by LeakyReLU(slope) (stylegan.py:33-34) in the same pass; slope = 1: AdaIN alone."""
        with torch.autocast("cuda", enabled=False):  # (N, z) x (z, C): fp32 costs nothing here
            gamma = self.style_w(w.float()) + 1
            beta = self.style_b(w.float())
        return GF.instance_norm_affine(to_act(x), gamma, beta, slope)

    def forward(self, x: Tensor, w: Tensor) -> Tensor:
        return self.fused(x, w, 1.0)


class LayerNorm2d(nn.LayerNorm):
    """modelling/common.py:19-24: nn.LayerNorm over the channel axis of an (N, C, H, W) tensor; channels-last makes that
    a row normalisation of each pixel."""

    def fused(self, x: Tensor, slope: float) -> Tensor:
        return GF.layer_norm2d(to_act(x), self.weight, self.bias, slope, self.eps)

    def forward(self, x: Tensor) -> Tensor:
        return self.fused(x, 1.0)


def get_norm(norm: str, dim: int):
    """modelling/common.py:27-33 ("instance" is only used by models outside this build's scope)."""
    return dict(none=nn.Identity, batch=partial(nn.BatchNorm2d, track_running_stats=False), layer=LayerNorm2d)[norm](dim)


def get_act(act: str):
    """modelling/common.py:36-43."""
    return dict(none=nn.Identity, relu=nn.ReLU, leaky_relu=partial(LeakyReLU, negative_slope=0.2), tanh=nn.Tanh)[act]()


def _fold_weight(weight: Tensor) -> Tensor:
    """(Co, C, 2m, 2m) -> (Co, 4C, m, m): the stride-2 kernel as a stride-1 kernel over space-to-depth input, input channel
    (dy*2+dx)*C + c and tap (by, bx) <- tap (2by+dy, 2bx+dx).  A permutation of the parameter, kept per parameter version
    (functional.FoldWeightFn)."""
    return GF.FoldWeightFn.apply(weight)


class _StridedConv2d(nn.Conv2d):
    """Conv2d(k, stride 2, p).  Even k with even padded size (DCGAN's 4x4 / 2 / 1, dcgan.py:21-29): space-to-depth + a
    stride-1 (k/2 x k/2) valid convolution -- every product the MFMA kernel forms is one the strided convolution needs.
    Otherwise: stride-1 correlation followed by keeping every second sample."""

    def forward(self, x: Tensor) -> Tensor:
        if self.stride == (1, 1):
            return Conv2d.forward(self, x)
        return self._forward_with(x, self.weight, self.kernel_size[0])

    def _forward_with(self, x: Tensor, weight: Tensor, k: int) -> Tensor:
        """The stride-2 layer with `weight` (Cout, Cin, k, k) in place of the parameter (a zero-extended odd kernel)."""
        if self.stride != (2, 2):
            raise GF._lib.GanAmdError("gan_amd strided Conv2d supports stride 1 or 2")
        x = to_act(x)
        p = self.padding[0]
        if k % 2 == 0 and weight.shape[2] == weight.shape[3] and p == self.padding[1] and (x.shape[2] + 2 * p) % 2 == 0 and (x.shape[3] + 2 * p) % 2 == 0:
            xs = GF.SpaceToDepthFn.apply(x, p)
            return GF.conv_bias_act(xs, _fold_weight(weight), self.bias, 0)
        y = GF.Conv2dFn.apply(x, self.weight, self.padding[0], self.padding[1])
        y = GF.upfirdn2d(y, torch.ones(1, 1, device=x.device), 1, 2, 0, 0, 0, 0)
        return GF.affine_act(y, b=self.bias) if self.bias is not None else y

    _check = Conv2d._check


class _ConvTranspose2d(nn.ConvTranspose2d):
    """ConvTranspose2d (stride 1 or 2) = zero insertion (upfirdn2d with a unit tap) + stride-1
    correlation with the transposed, flipped weights."""

    def forward(self, x: Tensor) -> Tensor:
        return self._forward_with(x, self.weight, self.kernel_size[0], self.output_padding[0])

    def _forward_with(self, x: Tensor, weight: Tensor, k: int, op: int) -> Tensor:
        """The layer with `weight` (Cin, Cout, k, k) and output_padding `op` in place of its own (a zero-extended odd kernel)."""
        p, st = self.padding[0], self.stride[0]
        if st not in (1, 2) or self.groups != 1 or self.dilation != (1, 1):
            raise GF._lib.GanAmdError("gan_amd ConvTranspose2d supports stride 1 or 2, groups 1, dilation 1")
        x = to_act(x)
        full = k - 1 - p
        if st == 2 and k % 2 == 0 and op == 0 and weight.shape[2] == weight.shape[3]:
            # the adjoint of the strided convolution above: input gradient of the folded valid convolution (a stride-1
            # correlation with the flipped, transposed folded weights), then depth-to-space with the crop
            n, _, h, w = x.shape
            m = k // 2
            y4 = GF.Conv2dDgradFn.apply(x, _fold_weight(weight), 0, 0)  # (N, 4 Cout, h + m - 1, w + m - 1)
            y = GF.DepthToSpaceFn.apply(y4, p, 2 * (h + m - 1) - 2 * p, 2 * (w + m - 1) - 2 * p)
            return GF.affine_act(y, b=self.bias) if self.bias is not None else y
        if st == 2:
            x = GF.upfirdn2d(x, torch.ones(1, 1, device=x.device), 2, 1, full, full - 1 + op, full, full - 1 + op)
            full = 0
        w = weight.flip(2, 3).transpose(0, 1)
        y = GF.Conv2dFn.apply(x, w, full, full)
        return GF.affine_act(y, b=self.bias) if self.bias is not None else y


def conv_norm_act(
    in_dim: int,
    out_dim: int,
    kernel_size: int,
    stride: int = 1,
    *,
    transpose: bool = False,
    padding_mode: str = "zeros",
    norm: str = "none",
    act: str = "none",
) -> nn.Sequential:
    """modelling/common.py:46-70."""
    conv_cls = _ConvTranspose2d if transpose else _StridedConv2d
    return nn.Sequential(
        conv_cls(in_dim, out_dim, kernel_size, stride, (kernel_size - 1) // 2, bias=norm in ("none", "layer"), padding_mode=padding_mode),
        get_norm(norm, out_dim),
        get_act(act),
    )
