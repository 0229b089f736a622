"""DCGAN generator / discriminator (BASELINE config 5: WGAN-GP, n_disc=5).

Reference: modelling/dcgan.py:16-72 -- same constructor signatures, child layout and state_dict keys.
Convolutions run on the MFMA kernels: the 4x4 stride-2 convolutions as space-to-depth (gan_space_to_depth) + a stride-1
2x2 valid convolution over the folded weight, the stride-2 transposed convolutions as that map's adjoint (input gradient
of the folded convolution + depth-to-space) -- no wasted products -- so the gradient-penalty double backward through
the discriminator stays on the HIP kernels.  BatchNorm (generator only, never double-differentiated in the
step) and ReLU/Tanh stay on stock PyTorch-ROCm, as SURVEY.md section 2 row 6 scopes them.
"""
from __future__ import annotations

import math

from torch import Tensor, nn

from .common import Conv2d, _ConvTranspose2d, conv_norm_act, get_norm, to_act


class DcGanDiscriminator(nn.Sequential):
    def __init__(self, img_size: int = 64, base_dim: int = 64, norm: str = "batch") -> None:
        super().__init__()
        self.append(conv_norm_act(3, base_dim, 4, 2, act="leaky_relu"))
        in_ch = base_dim
        depth = int(math.log2(img_size / 4))
        for _ in range(depth - 1):
            self.append(conv_norm_act(in_ch, in_ch * 2, 4, 2, norm=norm, act="leaky_relu"))
            in_ch *= 2
        self.append(Conv2d(in_ch, 1, 4))
        self.append(nn.Flatten(0))
        self.reset_parameters()

    def reset_parameters(self):
        self.apply(init_weights)

    def forward(self, imgs: Tensor) -> Tensor:
        x = to_act(imgs)
        for m in self:
            x = m(x)
        return x.float()


class DcGanGenerator(nn.Sequential):
    def __init__(self, img_size: int = 64, base_dim: int = 64, z_dim: int = 128, norm: str = "batch") -> None:
        depth = int(math.log2(img_size / 4))
        out_ch = base_dim * 2 ** (depth - 1)
        super().__init__()
        self.append(
            nn.Sequential(
                nn.Unflatten(-1, (z_dim, 1, 1)),
                _ConvTranspose2d(z_dim, out_ch, 4),
                get_norm(norm, out_ch),
                nn.ReLU(),
            )
        )
        in_ch = out_ch
        for _ in range(depth - 1):
            self.append(conv_norm_act(in_ch, in_ch // 2, 4, 2, transpose=True, norm=norm, act="relu"))
            in_ch //= 2
        self.append(conv_norm_act(in_ch, 3, 4, 2, transpose=True, act="tanh"))
        self.reset_parameters()

    def reset_parameters(self):
        self.apply(init_weights)


def init_weights(module: nn.Module):
    """N(0, 0.02) conv weights, N(1, 0.02) BatchNorm weights, zero biases (dcgan.py:65-72)."""
    if isinstance(module, nn.modules.conv._ConvNd):
        nn.init.normal_(module.weight, 0, 0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.modules.batchnorm._BatchNorm):
        nn.init.normal_(module.weight, 1, 0.02)
        nn.init.zeros_(module.bias)
