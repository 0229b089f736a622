"""Optimizers for the train step (reference: `getattr(torch.optim, cfg.optimizer)(model.parameters(), cfg.lr,
**cfg.optimizer_kwargs)`, train_celeba.py:75-77).

`make_optimizer` keeps that contract -- any torch.optim class by name -- and serves the README recipes' choice, Adam
(README.md:55,60; AdamW too), with `FlatAdam`: the same update rule as torch.optim.Adam run by ONE kernel over the
module's flat parameter / gradient arenas (gan_amd/utils.py: FlatArena) with the step count on the device, so the
whole optimizer step is two launches and can be captured in a HIP graph.  torch's own Adam needs capturable=True
for that, which costs a launch per parameter tensor for each bias-correction scalar (~600 launches per iteration
for the two StyleGAN2 networks)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import nn

from . import _lib
from . import functional as GF
from ._lib import check, ptr, stream_of
from .utils import FlatArena, arena_of

_FLAT_KEYS = {"betas", "eps", "weight_decay"}


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW (amsgrad = False, maximize = False) over a FlatArena.

    * one `param_group` holding the arena's parameters; `lr`, `betas`, `eps`, `weight_decay` are read from it at every
      step (schedulers that edit param_groups work; under HIP-graph capture the values at capture time are baked in);
    * `state[p]` exposes `step` / `exp_avg` / `exp_avg_sq` like torch's Adam -- the moments are views into two flat
      buffers, `step` is ONE device scalar shared by the arena (torch keeps one per tensor: the same number unless a
      parameter skips updates);
    * parameters whose `.grad` is None are skipped, as torch does."""

    def __init__(self, arena: FlatArena, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, decoupled: bool = False):
        if arena.grad is None:
            raise ValueError("FlatAdam needs an arena with a gradient buffer (arena_of(module, with_grad=True))")
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))
        super().__init__(arena.params, defaults)
        self.arena = arena
        self.decoupled = decoupled
        dev = arena.data.device
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self.step_t = torch.zeros(4, dtype=torch.float32, device=dev)  # [0] = updates done so far
        self._point_state()

    def _point_state(self) -> None:
        step = self.step_t[0]
        for p, o in zip(self.arena.params, self.arena.offsets):
            n = p.numel()
            self.state[p] = {"step": step, "exp_avg": self.exp_avg[o:o + n].view(p.shape), "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape)}

    def _runs(self) -> List[Tuple[int, int]]:
        """(offset, numel) ranges of the arena to update: everything when every parameter has a gradient in its arena
        slice (the train step's case); otherwise the gradients are gathered into the arena first and the parameters
        without one are left out."""
        arena = self.arena
        base = arena.grad.data_ptr()
        missing, stray = False, False
        for p, o in zip(arena.params, arena.offsets):
            if p.grad is None:
                missing = True
            elif p.grad.data_ptr() != base + 4 * o:
                stray = True
        if not missing and not stray:
            return [(0, arena.numel)]
        present = [p.grad is not None for p in arena.params]
        if stray:
            arena.gather_grads()  # re-attaches every slice; `present` remembers who really had a gradient
        runs: List[Tuple[int, int]] = []
        align = arena.ALIGN
        for p, o, ok in zip(arena.params, arena.offsets, present):
            if not ok:
                continue
            n = (p.numel() + align - 1) // align * align
            if runs and runs[-1][0] + runs[-1][1] == o:
                runs[-1] = (runs[-1][0], runs[-1][1] + n)
            else:
                runs.append((o, n))
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self.arena
        if not arena.intact():
            raise RuntimeError("FlatAdam: parameters were moved after flattening (model.to(...) / .half()); rebuild the optimizer")
        group = self.param_groups[0]
        lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
        lib = _lib.load()
        st = stream_of(arena.data)
        advance = 1
        for off, n in self._runs():
            check(lib.gan_adam_step(arena.data.data_ptr() + 4 * off, arena.grad.data_ptr() + 4 * off, self.exp_avg.data_ptr() + 4 * off,
                                    self.exp_avg_sq.data_ptr() + 4 * off, n, ptr(self.step_t), advance, float(lr), float(b1), float(b2),
                                    float(eps), float(wd), int(self.decoupled), st), "gan_adam_step")
            advance = 0
        # the kernel wrote through raw pointers: no parameter's version moved, so cached GEMM images of the old values
        # would still match their keys
        GF.invalidate_packed(arena.params)
        return loss

    def zero_grad(self, set_to_none: bool = False) -> None:
        """The gradients live in the arena: zero it in one launch and keep the slices attached."""
        self.arena.grad.zero_()
        self.arena.attach_grads()

    def load_state_dict(self, state_dict) -> None:
        """Accepts a torch.optim.Adam state_dict of the same parameters: the per-tensor moments are copied into the flat
        buffers (which stay the storage), the step count is taken from the first entry."""
        super().load_state_dict(state_dict)
        with torch.no_grad():
            first = True
            for p, o in zip(self.arena.params, self.arena.offsets):
                s = self.state.get(p)
                if not s:
                    continue
                n = p.numel()
                self.exp_avg[o:o + n].copy_(s["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
                if first:
                    self.step_t[0] = float(s["step"])
                    first = False
        self._point_state()


def make_optimizer(name: str, module: nn.Module, lr: float, kwargs: Dict, *, capturable: bool = False) -> torch.optim.Optimizer:
    """The optimizer `name` (a torch.optim class name, train_celeba.py:75) for `module`'s parameters.  Adam / AdamW with
    only betas / eps / weight_decay set run as FlatAdam on a ROCm device; everything else is torch's own class
    (with capturable=True where it exists and the step is going to be captured in a HIP graph)."""
    params = list(module.parameters())
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if name in ("Adam", "AdamW") and on_gpu and set(kwargs) <= _FLAT_KEYS:
        arena = arena_of(module, with_grad=True)
        kw = dict(kwargs)
        if name == "AdamW":
            kw.setdefault("weight_decay", 1e-2)  # torch.optim.AdamW's default
        return FlatAdam(arena, lr, decoupled=name == "AdamW", **kw)
    cls = getattr(torch.optim, name)
    kw = dict(kwargs)
    if capturable and name in ("Adam", "AdamW", "NAdam", "RAdam", "Adamax", "Adadelta", "Adagrad", "ASGD", "RMSprop", "Rprop"):
        kw.setdefault("capturable", True)
    return cls(params, lr, **kw)
