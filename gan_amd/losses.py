"""GAN objectives and gradient penalties (reference: losses.py; same registry keys and signatures).

The five objectives act on (N,) logit vectors -- a handful of scalar ops.  The two regularizers
are where the work is: they differentiate the discriminator w.r.t. its input with
create_graph=True, and the later loss.backward() differentiates that gradient again.  Every node
of both graphs is one of gan_amd.functional's Functions, so both orders run on the HIP kernels
(no autograd-of-autograd over PyTorch convolutions), and the squared-norm reductions over the
image-sized gradient use the channel-reduction kernel as well.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import functional as GF


def get_loss(name: str):
    return {
        "gan": GanLoss,
        "lsgan": LsGanLoss,
        "wgan": WganLoss,
        "hinge": HingeLoss,
        "relativistic-gan": RelativisticGanLoss,
    }[name]


def get_regularizer(name: str):
    return {"none": None, "wgan-gp": wgan_gp_regularizer, "r1": r1_regularizer}[name]


class GanLoss:  # losses.py:27-34
    @staticmethod
    def d_loss(d_reals: Tensor, d_fakes: Tensor) -> Tensor:
        return -F.logsigmoid(d_reals).mean() - F.logsigmoid(-d_fakes).mean()

    @staticmethod
    def g_loss(d_fakes: Tensor, d_reals_fn: Callable[[], Tensor]) -> Tensor:
        return -F.logsigmoid(d_fakes).mean()


class LsGanLoss:  # losses.py:38-45
    @staticmethod
    def d_loss(d_reals: Tensor, d_fakes: Tensor) -> Tensor:
        return (d_reals - 1).square().mean() + d_fakes.square().mean()

    @staticmethod
    def g_loss(d_fakes: Tensor, d_reals_fn: Callable[[], Tensor]) -> Tensor:
        return (d_fakes - 1).square().mean()


class WganLoss:  # losses.py:50-57
    @staticmethod
    def d_loss(d_reals: Tensor, d_fakes: Tensor) -> Tensor:
        return -d_reals.mean() + d_fakes.mean()

    @staticmethod
    def g_loss(d_fakes: Tensor, d_reals_fn: Callable[[], Tensor]) -> Tensor:
        return -d_fakes.mean()


class HingeLoss:  # losses.py:64-69
    @staticmethod
    def d_loss(d_reals: Tensor, d_fakes: Tensor) -> Tensor:
        return F.relu(1 - d_reals).mean() + F.relu(1 + d_fakes).mean()

    g_loss = WganLoss.g_loss


class RelativisticGanLoss:  # losses.py:74-83
    @staticmethod
    def d_loss(d_reals: Tensor, d_fakes: Tensor) -> Tensor:
        return -F.logsigmoid(d_reals - d_fakes).mean()

    @staticmethod
    def g_loss(d_fakes: Tensor, d_reals_fn: Callable[[], Tensor]) -> Tensor:
        with torch.no_grad():
            d_reals = d_reals_fn()
        return -F.logsigmoid(d_fakes - d_reals).mean()


def _sq_norm(g: Tensor, per_sample: bool) -> Tensor:
    """sum of squares over (C,H,W) per sample, or over everything: channel-reduction kernel, bilinear
    Function, so its backward (2*g*cotangent) is again a HIP kernel."""
    g = g.contiguous(memory_format=torch.channels_last)
    s = GF.ChanDotFn.apply(g, g, per_sample)
    return s.sum(dim=1) if per_sample else s.sum()


def wgan_gp_regularizer(disc: nn.Module, reals: Tensor, fakes: Tensor, d_reals: Tensor) -> Tensor:
    """losses.py:89-96: 10 * mean((||d D(inter)/d inter||_2 - 1)^2), inter = lerp(reals, fakes, U(0,1))."""
    alpha = torch.rand(fakes.shape[0], 1, 1, 1, device=fakes.device, dtype=fakes.dtype)
    inter = reals.detach().to(fakes.dtype).lerp(fakes.detach(), alpha).requires_grad_()
    d_inter = disc(inter)
    (d_grad,) = torch.autograd.grad(d_inter.sum(), inter, create_graph=True)
    d_grad_norm = _sq_norm(d_grad, True).sqrt()
    return (d_grad_norm - 1).square().mean() * 10


def r1_regularizer(disc: nn.Module, reals: Tensor, fakes: Tensor, d_reals: Tensor) -> Tensor:
    """losses.py:100-103: 5 * sum((d sum(D(reals)) / d reals)^2) / N.  `reals` must require grad."""
    (d_grad,) = torch.autograd.grad(d_reals.sum(), reals, create_graph=True)
    return _sq_norm(d_grad, False) / d_grad.shape[0] * 5
