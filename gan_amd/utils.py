"""Training utilities (reference: utils.py): CLI parser from a dataclass, image (un)normalisation,
prepare_model, data-loader plumbing, and EMA.

EMA differs in mechanism, not in result: the reference loops over parameters and launches one lerp_
per tensor (utils.py:95-96, ~100 tiny kernels for the StyleGAN2 generator); here both the live model's
and the EMA copy's parameters are views into two flat fp32 arenas, and one kernel (gan_ema_lerp)
updates all 10 M values.  The same arena is what the data-parallel gradient buckets slice
(gan_amd/distributed.py)."""
from __future__ import annotations

import argparse
import copy
import inspect
import json
from typing import Iterable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import functional as GF


def make_parser(fn):
    """utils.py:11-24: str/int/float -> typed flag, bool -> store_true, dict -> json.loads."""
    parser = argparse.ArgumentParser()
    for k, v in inspect.signature(fn).parameters.items():
        if v.annotation in (str, int, float):
            parser.add_argument(f"--{k}", type=v.annotation, default=v.default)
        elif v.annotation is bool:
            parser.add_argument(f"--{k}", action="store_true")
        elif v.annotation is dict:
            parser.add_argument(f"--{k}", type=json.loads, default=dict())
        else:
            raise RuntimeError(f"Unsupported type {v.annotation} for argument {k}")
    return parser


def normalize_img(x: Tensor) -> Tensor:
    return (x / 255 - 0.5) / 0.5


def unnormalize_img(x: Tensor) -> Tensor:
    return ((x * 0.5 + 0.5) * 255).round().clip(0, 255).to(torch.uint8)


def apply_spectral_norm(m: nn.Module):
    if isinstance(m, (nn.Linear, nn.modules.conv._ConvNd, nn.Embedding)):
        nn.utils.parametrizations.spectral_norm(m)


def prepare_model(model: nn.Module, *, spectral_norm: bool = False, channels_last: bool = False, compile: bool = False) -> None:
    """utils.py:40-48.  The kernels are channels-last internally whatever this flag says (it still
    converts 4-D parameters, as the reference does).  `spectral_norm` applies apply_spectral_norm (the
    reference passes the bool itself to model.apply -- utils.py:44, a bug off the StyleGAN2 path).
    `compile` is rejected: there is no tracing compiler in this build (HIP kernels + hipGraph instead)."""
    if spectral_norm:
        model.apply(apply_spectral_norm)
    if channels_last:
        model.to(memory_format=torch.channels_last)
    if compile:
        raise RuntimeError("gan_amd does not use torch.compile; drop --compile")


def _tensor_to(x, *, device=None, channels_last: bool = False):
    if isinstance(x, Tensor):
        if channels_last and x.ndim == 4:
            return x.to(device=device, memory_format=torch.channels_last)
        return x.to(device=device)
    if isinstance(x, (tuple, list)):
        return x.__class__(_tensor_to(item, device=device, channels_last=channels_last) for item in x)
    raise ValueError(f"Unuspported type {type(x)}")


def cycle(iterator, *, device=None, channels_last: bool = False):
    while True:
        for x in iterator:
            yield _tensor_to(x, device=device, channels_last=channels_last)


def prepare_train_dloader(ds, batch_size: int, *, device=None, channels_last: bool = False):
    from torch.utils.data import DataLoader

    dloader = DataLoader(ds, batch_size, shuffle=True, num_workers=4, pin_memory=True, drop_last=True)
    return cycle(dloader, device=device, channels_last=channels_last)


# ----------------------------------------------------------------------------------------------
# flat parameter arenas
# ----------------------------------------------------------------------------------------------
class FlatArena:
    """All floating-point parameters of a module re-pointed at slices of ONE fp32 buffer (16-byte
    aligned slices, parameter order preserved) and, optionally, their .grad at slices of a second one."""

    ALIGN = 4  # elements (16 bytes)

    def __init__(self, params: Iterable[nn.Parameter], with_grad: bool = False):
        self.params: List[nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FlatArena: no parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("FlatArena: parameters must be fp32 on one device")
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad: Optional[Tensor] = torch.zeros(off, dtype=torch.float32, device=dev) if with_grad else None
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o : o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        if with_grad:
            self.attach_grads()

    def attach_grads(self) -> None:
        """(Re)point every p.grad at its slice of the gradient arena (autograd then accumulates in place)."""
        for p, o in zip(self.params, self.offsets):
            p.grad = self.grad[o : o + p.numel()].view(p.shape)

    def detach_grads(self) -> None:
        """p.grad = None: the next backward pass hands its gradient tensors over instead of adding them into
        the arena slices one launch per parameter; gather_grads() then files them with one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def begin_direct(self, select=lambda p: p.dim() == 4 or getattr(p, "_gan_direct", False)) -> None:
        """Before a backward pass run under functional.direct_param_grads(): the parameters `select` picks (conv
        weights: the tensors whose gradients the wgrad kernels can add in place; biases that functional._bias_grad_for
        has marked as summed by the channel-sum kernel) get their ZEROED arena slices as
        .grad, every other parameter None (autograd then hands its first gradient over for free); gather_grads()
        files the rest afterwards."""
        views = []
        for p, o in zip(self.params, self.offsets):
            if p.requires_grad and select(p):
                p.grad = self.grad[o : o + p.numel()].view(p.shape)
                views.append(p.grad)
            else:
                p.grad = None
        if views:
            torch._foreach_zero_(views)

    def gather_grads(self) -> None:
        views, srcs, missing = [], [], []
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None:
                if p.requires_grad:
                    missing.append(self.grad[o : o + p.numel()])
                continue
            if g.data_ptr() != self.grad.data_ptr() + 4 * o:
                views.append(self.grad[o : o + p.numel()].view(p.shape))
                srcs.append(g)
        if missing:
            # a parameter that took no part in this pass: its slice must read zero.  Only ITS slice: the slices handed
            # out by begin_direct() already hold the sums the kernels added in place during the pass
            torch._foreach_zero_(missing)
        if views:
            torch._foreach_copy_(views, srcs)
        self.attach_grads()

    def intact(self) -> bool:
        """False once someone replaced p.data (model.to(...), .half(), load with assign=True)."""
        base = self.data.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def slice_of(self, index: int) -> Tuple[int, int]:
        return self.offsets[index], self.params[index].numel()


def arena_of(module: nn.Module, with_grad: bool = False) -> FlatArena:
    """Flat arena of a module's parameters, created once and cached on the module."""
    arena: Optional[FlatArena] = getattr(module, "_gan_amd_arena", None)
    if arena is None or not arena.intact() or (with_grad and arena.grad is None):
        arena = FlatArena(module.parameters(), with_grad=with_grad)
        object.__setattr__(module, "_gan_amd_arena", arena)
    return arena


class EMA(nn.Module):
    """utils.py:75-99: deepcopy; counter; no-op before `warmup`, copy at `warmup`, then
    ema += (p - ema) * (1 - beta) over PARAMETERS only (buffers are not averaged)."""

    def __init__(self, model: nn.Module, beta: float = 0.999, warmup: int = 100) -> None:
        super().__init__()
        stash = model.__dict__.pop("_gan_amd_arena", None)  # arenas are rebuilt, not deep-copied
        self.ema_model = copy.deepcopy(model)
        if stash is not None:
            object.__setattr__(model, "_gan_amd_arena", stash)
        self.model = [model]
        self.beta = beta
        self.warmup = warmup
        self.register_buffer("counter", torch.tensor(0, dtype=torch.long))
        self._count = 0  # host mirror of `counter`: no device sync in step()

    @torch.no_grad()
    def step(self) -> None:
        self.counter += 1
        self._count += 1
        if self._count < self.warmup:
            return
        src = arena_of(self.model[0])
        dst = arena_of(self.ema_model)
        if self._count == self.warmup:
            dst.data.copy_(src.data)
        else:
            GF.raw_ema_lerp(dst.data, src.data, 1 - self.beta)
        # both writes go through the arena: no parameter's version moves, so the packed weight images that
        # gen_ema(z) cached at an earlier step would still match.  Drop them.
        GF.invalidate_packed(dst.params)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._count = int(self.counter.item())
        return out

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
