"""The StyleGAN2 discriminator (modelling/__init__.py:37 maps "stylegan2" to ProgressiveGanDiscriminator).

Reference: modelling/progressive_gan.py:20-82,136-140 -- same class names, constructor signatures,
child-module layout and state_dict keys.  Every convolution, leaky-ReLU, pooling and residual op runs
on the HIP kernels through Functions whose backward is again made of those kernels, so the R1 /
WGAN-GP double backward (losses.py:89-103) never differentiates through PyTorch ops' autograd.

Fusions per block (progressive_gan.py:43-54):  each LeakyReLU is applied by the convolution that
consumes it while it stages its input, AvgPool2d(2) is applied to the MFMA accumulators of the
convolution that precedes it (the shortcut's pool -> 1x1 conv runs as conv -> pool, the same linear
map), and the adjoints of both ride in the input-/weight-gradient kernels; conv biases that precede
AvgPool2d are added after it (avg_pool(z + b) = avg_pool(z) + b) on the same accumulators; the residual
    out = shortcut + layers * scale
rides on the last convolution (scale folded into its parameters, shortcut as its residual input).
LeakyReLU is applied out of place (the shipped inplace=True breaks autograd).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .. import functional as GF
from .common import AvgPool2d, Conv2d, LeakyReLU, to_act


class MinibatchStdDev(nn.Module):
    def __init__(self, group_size: int = 4) -> None:
        super().__init__()
        self.group_size = group_size

    def forward(self, imgs: Tensor) -> Tensor:
        """(N, C, H, W) -> (N, C+1, H, W); groups are {j + k*N/G} (progressive_gan.py:27), population
        variance, eps 1e-6 inside the sqrt, mean over (C, H, W)."""
        imgs = to_act(imgs)
        n, c, h, w = imgs.shape
        if n % self.group_size != 0:
            raise GF._lib.GanAmdError(f"MinibatchStdDev: batch {n} is not a multiple of group_size {self.group_size}")
        return GF.minibatch_stddev(imgs, self.group_size)  # statistic + the appended plane: two launches


class ProgressiveGanDiscriminatorBlock(nn.Module):
    """blur_size (default None = the reference's block): both AvgPool2d(2) become `Blur(blur_size, down=2)` FIR
    down-samplers, the pairing of blur_conv_down (nvidia_ops.py:104-123) -- SURVEY.md 8 f1.  AvgPool2d(2) IS
    Blur(2, down=2) (taps [[1,1],[1,1]]/4, pads (0,1)), so blur_size=2 reproduces the reference block exactly."""

    def __init__(self, in_dim: int, out_dim: int, blur_size=None) -> None:
        super().__init__()
        if blur_size is None:
            down = lambda: AvgPool2d(2)  # noqa: E731
        else:
            from .nvidia_ops import Blur

            down = lambda: Blur(blur_size, down=2)  # noqa: E731
        self.layers = nn.Sequential(
            LeakyReLU(0.2),
            Conv2d(in_dim, in_dim, 3, 1, 1),
            LeakyReLU(0.2),
            Conv2d(in_dim, out_dim, 3, 1, 1),
            down(),
        )
        self.shortcut = nn.Sequential(down(), Conv2d(in_dim, out_dim, 1))
        self.scale = nn.Parameter(torch.full((out_dim, 1, 1), 1e-4))

    def forward(self, x: Tensor) -> Tensor:
        x = to_act(x)
        act0, conv1, act1, conv2, down = self.layers
        conv_s = self.shortcut[1]
        if not isinstance(down, AvgPool2d):
            # FIR down-sampling: activation + convolution + bias fused per launch and the two FIR passes.  The FIR is linear
            # and per channel, so the block's scale folds into conv2's parameters as it does below, and the merge
            # shortcut + layers * scale (progressive_gan.py:54) rides as the residual of the shortcut's 1x1 convolution:
            # no pass over the activations for the scale or the sum, forward or backward
            h = GF.conv_bias_act(x, conv1.weight, conv1.bias, conv1.padding[0], 1.0, act0.negative_slope)
            if conv2.weight.dtype == torch.float32 and self.scale.dtype == torch.float32:
                w2s, b2s = GF.scaled_conv_params(conv2.weight, conv2.bias, self.scale)
                h = down(GF.conv_bias_act(h, w2s, b2s, conv2.padding[0], 1.0, act1.negative_slope))
                return GF.conv_bias_act(self.shortcut[0](x), conv_s.weight, conv_s.bias, 0, 1.0, 1.0, False, h)
            h = down(GF.conv_bias_act(h, conv2.weight, conv2.bias, conv2.padding[0], 1.0, act1.negative_slope))
            s = conv_s(self.shortcut[0](x))
            return GF.affine_act(h, a=self.scale.reshape(-1), r=s)
        # act -> conv1 (+bias) and the shortcut (pool -> conv_s; for thin layers conv_s -> pool, the same linear
        # map, in one launch) share x: one autograd node, so their input gradients come out of one kernel
        if GF._FUSE_DBLOCK and conv2.weight.dtype == torch.float32 and self.scale.dtype == torch.float32 and conv1.kernel_size == (3, 3):
            # the whole block as one autograd node: its input gradient's own backward (R1's second order) is one chain
            w2s, b2s = GF.scaled_conv_params(conv2.weight, conv2.bias, self.scale)
            return GF.DBlockFn.apply(x, conv1.weight, conv1.bias, conv_s.weight, conv_s.bias, w2s, b2s, act0.negative_slope,
                                     act1.negative_slope, conv_s.out_channels <= 64)
        h, s = GF.DBlockInputFn.apply(x, conv1.weight, conv1.bias, conv_s.weight, act0.negative_slope, conv_s.out_channels <= 64,
                                      conv_s.bias)
        # act -> conv2 -> pool (+ bias: avg_pool(z + b) = avg_pool(z) + b) in one launch, and the residual merge
        # shortcut + layers * scale (progressive_gan.py:54) with it: the scale is folded into conv2's parameters
        # (conv(x, W) * scale + b * scale = conv(x, W * scale) + b * scale, weight-sized work once per version of the
        # parameters) and the shortcut rides as the convolution's residual -- no pass over the activations, forward or
        # backward, first or second order
        if conv2.weight.dtype == torch.float32 and self.scale.dtype == torch.float32:
            w2s, b2s = GF.scaled_conv_params(conv2.weight, conv2.bias, self.scale)
            return GF.conv_bias_act(h, w2s, b2s, conv2.padding[0], 1.0, act1.negative_slope, True, s)
        h = GF.conv_bias_act(h, conv2.weight, conv2.bias, conv2.padding[0], 1.0, act1.negative_slope, True)
        return GF.affine_act(h, a=self.scale.reshape(-1), r=s)


class ProgressiveGanDiscriminator(nn.Sequential):
    def __init__(self, img_size: int, img_channels: int = 3, base_dim: int = 16, blur_size=None) -> None:
        super().__init__()
        depth = int(math.log2(img_size // 4))
        self.append(Conv2d(img_channels, base_dim, 1))
        in_ch = base_dim

        for _ in range(depth):
            out_ch = min(in_ch * 2, 512)
            self.append(ProgressiveGanDiscriminatorBlock(in_ch, out_ch, blur_size))
            in_ch = out_ch

        self.append(
            nn.Sequential(
                MinibatchStdDev(),
                Conv2d(in_ch + 1, in_ch, 3, 1, 1),
                LeakyReLU(0.2),
                Conv2d(in_ch, 1, 4),
                nn.Flatten(0),
            )
        )
        self.reset_parameters()

    def reset_parameters(self):
        self.apply(init_weights)

    def forward(self, imgs: Tensor) -> Tensor:
        if imgs.dim() != 4:
            raise GF._lib.GanAmdError(f"expected a 4-D (N, C, H, W) tensor, got shape {tuple(imgs.shape)}")
        x = imgs  # the first convolution reads the image as it is (any strides, fp32 or bf16)
        *body, head = list(self)
        for m in body:
            x = m(x)
        mbstd, conv1, act, conv2, flat = head
        x = mbstd(x)
        x = GF.conv_bias_act(x, conv1.weight, conv1.bias, conv1.padding[0], act.negative_slope)
        x = conv2(x)
        return flat(x).float()


class ProgressiveGanGeneratorBlock(nn.Module):
    """progressive_gan.py:85-100: shortcut(x) + layers(x) * scale with layers = LayerNorm2d -> LeakyReLU -> Upsample ->
    Conv3x3 -> LeakyReLU -> Conv3x3 and shortcut = Conv1x1 -> Upsample.  Launches: LayerNorm2d + LeakyReLU; Upsample +
    Conv (nearest 2x in the gather: leaky-ReLU and nearest upsampling commute); LeakyReLU + Conv; Conv1x1; scale; skip."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        from .common import LayerNorm2d, Upsample

        self.layers = nn.Sequential(
            LayerNorm2d(in_dim),
            LeakyReLU(0.2),
            Upsample(scale_factor=2.0),
            Conv2d(in_dim, out_dim, 3, 1, 1),
            LeakyReLU(0.2),
            Conv2d(out_dim, out_dim, 3, 1, 1),
        )
        self.shortcut = nn.Sequential(Conv2d(in_dim, out_dim, 1), Upsample(scale_factor=2.0))
        self.scale = nn.Parameter(torch.full((out_dim, 1, 1), 1e-4))

    def forward(self, x: Tensor) -> Tensor:
        from .stylegan2 import ModConvFn

        x = to_act(x)
        norm, act0, _, conv1, act1, conv2 = self.layers
        h = norm.fused(x, act0.negative_slope)
        h = ModConvFn.apply(h, conv1.weight, None, None, conv1.bias, None, None, 1.0, True)
        h = GF.conv_bias_act(h, conv2.weight, conv2.bias, conv2.padding[0], 1.0, act1.negative_slope)
        s = self.shortcut[0](x)
        return GF.up2_add(s, GF.affine_act(h, a=self.scale.reshape(-1)))


class ProgressiveGanGenerator(nn.Sequential):
    """progressive_gan.py:103-133 (same child layout and state_dict keys)."""

    def __init__(self, img_size: int, img_channels: int = 3, z_dim: int = 128, base_dim: int = 16) -> None:
        super().__init__()
        from .common import LayerNorm2d, _ConvTranspose2d

        out_ch = min(base_dim * img_size // 4, 512)
        self.append(nn.Sequential(nn.LayerNorm(z_dim), nn.Unflatten(-1, (-1, 1, 1)), _ConvTranspose2d(z_dim, out_ch, 4)))
        in_ch = out_ch

        depth = int(math.log2(img_size // 4))
        for i in range(depth):
            out_ch = min(base_dim * img_size // 4 // 2 ** (i + 1), 512)
            self.append(ProgressiveGanGeneratorBlock(in_ch, out_ch))
            in_ch = out_ch

        self.append(nn.Sequential(LayerNorm2d(in_ch), LeakyReLU(0.2), Conv2d(in_ch, img_channels, 1)))
        self.reset_parameters()

    def reset_parameters(self):
        self.apply(init_weights)

    def forward(self, z: Tensor) -> Tensor:
        first, *blocks, last = list(self)
        with torch.autocast("cuda", enabled=False):
            zn = first[0](z.float())
        x = first[2](first[1](zn).to(GF.act_dtype(z)))
        for blk in blocks:
            x = blk(x)
        norm, act, conv = last
        return conv(norm.fused(x, act.negative_slope))


def init_weights(module: nn.Module):
    """kaiming-normal weights, zero biases on every conv / linear (progressive_gan.py:136-140)."""
    if isinstance(module, (nn.modules.conv._ConvNd, nn.Linear)):
        nn.init.kaiming_normal_(module.weight)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
