"""DiffAugment (reference diff_augment.py, https://arxiv.org/abs/2006.10738): the same names and call signatures --
rand_brightness / rand_saturation / rand_contrast, ColorJitter, translate, RandomTranslate, RandomCutout, DiffAugment
(train_celeba.py:63 wraps the discriminator in nn.Sequential(DiffAugment(), disc)).

The reference runs the three jitters, the pad and the slice-assign as eight kernels.  Here every piece, and the whole
DiffAugment pipeline, is ONE call of gan_diffaug (gan_amd/csrc/diffaug.hip): the pipeline is an affine map of the image
batch, its gradient the adjoint map and the gradient of that the linear part again, so `DiffAugFn` is closed under
differentiation and R1 (which differentiates the augmented discriminator twice) stays on the kernel.

Random draws: the per-sample jitters are drawn exactly as the reference draws them (torch.rand(N, 1, 1, 1) in the input's
dtype on its device, brightness -> saturation -> contrast).  The standalone RandomTranslate / RandomCutout keep the
reference's host-side `torch.randint(...).item()`; DiffAugment itself draws the four integers on the device so that the
augmented pass has no host synchronisation and can be captured in a HIP graph (train_celeba.py's graph path)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import _lib
from .functional import Function
from ._lib import check, dtype_code, ptr, require_gpu, stream_of

def rand_int(lo: int, hi: int) -> int:
    return torch.randint(lo, hi, ()).item()


def _raw_diffaug(x: Tensor, usk: Tensor, shift: Tensor, mode: int) -> Tensor:
    """gan_diffaug on x (N, C <= 4, H, W; any strides; fp32 / bf16) -> channels-last tensor of the same shape and dtype."""
    require_gpu(x, usk, shift)
    lib = _lib.load()
    if x.dim() != 4 or x.shape[1] > 4:
        raise _lib.GanAmdError(f"diff_augment: expects an image batch (N, C <= 4, H, W), got {tuple(x.shape)}")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.detach()
    n, c, h, w = x.shape
    usk = usk.detach().to(torch.float32).reshape(n, 3).contiguous()
    shift = shift.detach().to(torch.int32).reshape(4).contiguous()
    out = torch.empty((n, c, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws_floats = lib.gan_diffaug_workspace(n)
    ws = torch.empty((max(ws_floats, 1),), dtype=torch.float32, device=x.device)
    sn, sc, sh, sw = x.stride()
    check(lib.gan_diffaug(ptr(x), sn, sc, sh, sw, ptr(out), ptr(usk), ptr(shift), mode, n, c, h, w, ptr(ws), ws_floats, dtype_code(x), stream_of(x)),
          "gan_diffaug")
    return out


class DiffAugFn(Function):
    """mode 0: y = A x + c (the augmentation), 1: y = A x, 2: y = A' x.  backward(0 | 1) = mode 2, backward(2) = mode 1."""

    @staticmethod
    def forward(ctx, x, usk, shift, mode):
        ctx.save_for_backward(usk, shift)
        ctx.mode = mode
        ctx.x_dtype = x.dtype
        return _raw_diffaug(x, usk, shift, mode)

    @staticmethod
    def backward(ctx, g):
        usk, shift = ctx.saved_tensors
        gx = DiffAugFn.apply(g, usk, shift, 1 if ctx.mode == 2 else 2).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        return gx, None, None, None


def diff_augment(x: Tensor, brightness: Tensor, saturation: Tensor, contrast: Tensor, translate_x, translate_y, offset_y, offset_x) -> Tensor:
    """The whole pipeline with explicit draws: `brightness` / `saturation` / `contrast` are the torch.rand values of
    diff_augment.py:14,19,24 (N values each), the four integers those of :42,48-49 (Python ints or 0-d tensors)."""
    n = x.shape[0]
    usk = torch.stack([t.reshape(n).float() for t in (brightness, saturation, contrast)], dim=1)
    ints = [t.to(device=x.device, dtype=torch.int32).reshape(()) if isinstance(t, Tensor) else torch.tensor(int(t), dtype=torch.int32, device=x.device)
            for t in (translate_x, translate_y, offset_y, offset_x)]
    return DiffAugFn.apply(x, usk, torch.stack(ints), 0)


def _identity(x: Tensor):
    """Jitter draws that leave the image unchanged (u_b = 0.5, S = 2 * 0.5, K = 0.5 + 0.5) and a shift that moves and cuts
    nothing (the cut-out square parked outside the image)."""
    n = x.shape[0]
    usk = torch.full((n, 3), 0.5, dtype=torch.float32, device=x.device)
    shift = torch.tensor([0, 0, 2 * x.shape[-2], 2 * x.shape[-1]], dtype=torch.int32, device=x.device)
    return usk, shift


def _draw(x: Tensor) -> Tensor:
    return torch.rand(*x.shape[:-3], 1, 1, 1, dtype=x.dtype, device=x.device)


def rand_brightness(x: Tensor) -> Tensor:
    usk, shift = _identity(x)
    usk[:, 0] = _draw(x).reshape(-1)
    return DiffAugFn.apply(x, usk, shift, 0)


def rand_saturation(x: Tensor) -> Tensor:
    usk, shift = _identity(x)
    usk[:, 1] = _draw(x).reshape(-1)
    return DiffAugFn.apply(x, usk, shift, 0)


def rand_contrast(x: Tensor) -> Tensor:
    usk, shift = _identity(x)
    usk[:, 2] = _draw(x).reshape(-1)
    return DiffAugFn.apply(x, usk, shift, 0)


class ColorJitter(nn.Module):
    def forward(self, x: Tensor):
        usk, shift = _identity(x)
        for j in range(3):  # brightness, saturation, contrast: the reference's order of draws
            usk[:, j] = _draw(x).reshape(-1)
        return DiffAugFn.apply(x, usk, shift, 0)


def translate(x: Tensor, translate_x: int, translate_y: int) -> Tensor:
    usk, shift = _identity(x)
    shift[0], shift[1] = int(translate_x), int(translate_y)
    return DiffAugFn.apply(x, usk, shift, 0)


class RandomTranslate(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        H, W = x.shape[-2:]
        return translate(x, rand_int(-(W // 8), W // 8 + 1), rand_int(-(H // 8), H // 8 + 1))


class RandomCutout(nn.Module):
    """The reference zeroes the square in its input and returns that tensor (in DiffAugment the input is RandomTranslate's
    fresh output, so nobody else sees the write); this returns a new tensor and leaves the input as it was."""

    def forward(self, x: Tensor) -> Tensor:
        H, W = x.shape[-2:]
        usk, shift = _identity(x)
        shift[2], shift[3] = rand_int(0, H // 2), rand_int(0, W // 2)  # offset_y, then offset_x: the reference's order
        return DiffAugFn.apply(x, usk, shift, 0)


class DiffAugment(nn.Sequential):
    """nn.Sequential(ColorJitter(), RandomTranslate(), RandomCutout()) as the reference builds it (the children are there
    for introspection and state_dict parity -- there is no state); forward runs the three as one kernel call."""

    def __init__(self) -> None:
        super().__init__(
            ColorJitter(),
            RandomTranslate(),
            RandomCutout(),
        )

    def forward(self, x: Tensor) -> Tensor:
        n = x.shape[0]
        H, W = x.shape[-2:]
        usk = torch.stack([_draw(x).reshape(n).float() for _ in range(3)], dim=1)
        dev = x.device
        # the four integers of :42 and :48-49, drawn on the device (no .item(): the pass stays capturable)
        shift = torch.stack([
            torch.randint(-(W // 8), W // 8 + 1, (), device=dev),
            torch.randint(-(H // 8), H // 8 + 1, (), device=dev),
            torch.randint(0, max(H // 2, 1), (), device=dev),
            torch.randint(0, max(W // 2, 1), (), device=dev),
        ]).to(torch.int32)
        return DiffAugFn.apply(x, usk, shift, 0)
