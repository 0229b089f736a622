"""Differentiable ops over the HIP kernels of libgan_amd.so.

Layout: every activation is a 4-D tensor of logical shape (N, C, H, W) -- the shape the
reference's modules see -- stored channels-last (C fastest), fp32 or bf16.

Differentiation: each `torch.autograd.Function` below expresses its backward through other
Functions of this file (the pattern of the reference's UpFIRDn2d, modelling/nvidia_ops.py:40-54),
so gradients of any order -- R1 and WGAN-GP need second order through the discriminator
(losses.py:89-103) -- run on the same HIP kernels and never fall back to autograd-of-autograd over
PyTorch ops.  The primitive set is closed under differentiation:

    conv2d  <-> conv2d_dgrad <-> conv2d_wgrad          (a convolution is bilinear in x and W)
    upfirdn2d <-> upfirdn2d (flipped taps, up/down swapped)
    chan_scale <-> chan_dot                             (x * a[n,c]  and  sum_hw A*B)
    mask_mul(g, y) = g * lrelu'(y)                      (linear in g; d/dy = 0 almost everywhere)

Fused forward kernels (affine_act, modulated conv) are Functions whose backward is composed of
these primitives.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.autograd import Function as _TorchFunction

from . import _lib
from ._lib import check, dtype_code, ptr, require_gpu, stream_of

CL = torch.channels_last


# ======================================================================================
# optional per-launch measurement (bench.py's roofline leg): HIP events on the launch stream
# ======================================================================================
class KernelProfile:
    """Brackets every C-ABI launch with two events on the stream the kernel is enqueued on and keeps its
    algorithmic work (flops for the MFMA kernels, HBM bytes for the streaming kernels).  Off by default:
    the training path pays one `is None` test per launch."""

    def __init__(self):
        self.records = []  # (family, flops, bytes, start_event, end_event)

    def launch(self, family: str, flops: float, nbytes: float, thunk, tag=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = thunk()
        e1.record()
        self.records.append((family, float(flops), float(nbytes), e0, e1, tag))
        return out

    def by_shape(self):
        """(family, tag) -> dict(launches, ms, flops, bytes), for the per-layer table of DESIGN.md."""
        out = {}
        for family, flops, nbytes, e0, e1, tag in self.records:
            d = out.setdefault((family, tag), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out

    def summary(self):
        """family -> dict(launches, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for family, flops, nbytes, e0, e1, _tag in self.records:
            d = out.setdefault(family, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


_PROFILE: Optional[KernelProfile] = None


def set_profile(profile: Optional[KernelProfile]) -> None:
    global _PROFILE
    _PROFILE = profile


def _run(family: str, flops: float, nbytes: float, thunk, tag=None):
    if _PROFILE is None:
        return thunk()
    return _PROFILE.launch(family, flops, nbytes, thunk, tag)


def _nbytes(*tensors) -> int:
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


# ======================================================================================
# raw (non-differentiable) kernel launchers
# ======================================================================================
def as_cl(x: Tensor) -> Tensor:
    """4-D tensor, dense, channels-last."""
    if x.dim() != 4:
        raise _lib.GanAmdError(f"expected a 4-D (N, C, H, W) tensor, got shape {tuple(x.shape)}")
    return x.contiguous(memory_format=CL)


def _empty_like_cl(x: Tensor, shape) -> Tensor:
    return torch.empty(shape, dtype=x.dtype, device=x.device, memory_format=CL)


def _f32(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def raw_upfirdn2d(x: Tensor, taps: Tensor, up: int, down: int, px1: int, py1: int, ho: int, wo: int) -> Tensor:
    require_gpu(x, taps)
    lib = _lib.load()
    x = as_cl(x.detach())
    taps32 = _f32(taps)
    n, c, h, w = x.shape
    ky, kx = taps32.shape
    y = _empty_like_cl(x, (n, c, ho, wo))
    check(
        _run(
            "upfirdn2d", 0, _nbytes(x, y),
            lambda: lib.gan_upfirdn2d(ptr(x), ptr(taps32), ptr(y), n, h, w, c, ky, kx, up, down, px1, py1, ho, wo, dtype_code(x), stream_of(x)),
            (n, c, h, w, ky, up, down),
        ),
        "gan_upfirdn2d",
    )
    return y


def _side(t: Optional[Tensor], n: int, c: int, what: str) -> Tuple[Optional[Tensor], int]:
    """fp32 side vector: (C,) -> stride 0, (N, C) -> stride C."""
    if t is None:
        return None, 0
    t = _f32(t)
    if t.numel() == c:
        return t.reshape(c), 0
    if t.numel() == n * c:
        return t.reshape(n, c), c
    raise _lib.GanAmdError(f"{what}: expected {c} or {n}x{c} values, got shape {tuple(t.shape)}")


def raw_affine_act(
    x: Tensor,
    a: Optional[Tensor] = None,
    b: Optional[Tensor] = None,
    noise: Optional[Tensor] = None,
    ns: Optional[Tensor] = None,
    r: Optional[Tensor] = None,
    slope: float = 1.0,
) -> Tensor:
    require_gpu(x, a, b, noise, ns, r)
    lib = _lib.load()
    x = as_cl(x.detach())
    n, c, h, w = x.shape
    a32, a_sn = _side(a, n, c, "affine_act a")
    b32, _ = _side(b, 1, c, "affine_act b")
    ns32, _ = _side(ns, 1, c, "affine_act noise scale")
    if noise is not None:
        noise = noise.detach().to(x.dtype).reshape(n, h * w).contiguous()
    if r is not None:
        r = as_cl(r.detach().to(x.dtype))
        assert r.shape == x.shape
    y = _empty_like_cl(x, x.shape)
    check(
        _run(
            "affine_act", 0, _nbytes(x, y, r, noise),
            lambda: lib.gan_affine_act_fwd(
                ptr(x), ptr(a32), a_sn, ptr(b32), ptr(noise), ptr(ns32), ptr(r), ptr(y), n, h * w, c, float(slope), dtype_code(x), stream_of(x)
            ),
            (n, c, h, w),
        ),
        "gan_affine_act_fwd",
    )
    return y


def raw_mask_scale(g: Tensor, ymask: Optional[Tensor], a: Optional[Tensor], slope: float) -> Tensor:
    require_gpu(g, ymask, a)
    lib = _lib.load()
    g = as_cl(g.detach())
    n, c, h, w = g.shape
    if ymask is not None:
        ymask = as_cl(ymask.detach().to(g.dtype))
        assert ymask.shape == g.shape
    a32, a_sn = _side(a, n, c, "mask_scale a")
    out = _empty_like_cl(g, g.shape)
    check(
        _run(
            "mask_scale", 0, _nbytes(g, ymask, out),
            lambda: lib.gan_mask_scale(ptr(g), ptr(ymask), ptr(a32), a_sn, ptr(out), n, h * w, c, float(slope), dtype_code(g), stream_of(g)),
            (n, c, h, w),
        ),
        "gan_mask_scale",
    )
    return out


def raw_chan_dot(A: Tensor, B: Optional[Tensor], b_mode: int, ymask: Optional[Tensor], slope: float, per_sample: bool,
                 into: Optional[Tensor] = None) -> Tensor:
    """fp32 (N, C) if per_sample else (C,).  `into`: an fp32 buffer of that many elements the sums are ADDED to."""
    require_gpu(A, B, ymask)
    lib = _lib.load()
    A = as_cl(A.detach())
    n, c, h, w = A.shape
    if b_mode == 1:
        B = as_cl(B.detach().to(A.dtype))
        assert B.shape == A.shape
    elif b_mode == 2:
        B = B.detach().to(A.dtype).reshape(n, h * w).contiguous()
    else:
        B = None
    if ymask is not None:
        ymask = as_cl(ymask.detach().to(A.dtype))
    # two deterministic stages in the library (partial rows -> fold): the batch sum comes out of the same call
    if into is not None:
        assert into.dtype == torch.float32 and into.is_contiguous() and into.numel() == (n * c if per_sample else c)
    out = into if into is not None else torch.empty((n, c) if per_sample else (c,), dtype=torch.float32, device=A.device)
    ws_floats = lib.gan_chan_dot_workspace(n, h * w, c, dtype_code(A))
    ws = torch.empty((max(ws_floats, 1),), dtype=torch.float32, device=A.device)
    check(
        _run(
            "chan_dot", 0, _nbytes(A, B, ymask),
            lambda: lib.gan_chan_dot(ptr(A), ptr(B), b_mode, ptr(ymask), float(slope), ptr(out), c if per_sample else 0, n, h * w, c, ptr(ws), ws_floats,
                                     int(into is not None), dtype_code(A), stream_of(A)),
            (n, c, h, w, b_mode, int(ymask is not None), int(per_sample)),
        ),
        "gan_chan_dot",
    )
    return out


# ---- all Noise draws of a generator pass from one launch -----------------------------------------------------------
# The reference draws torch.randn(N, 1, H, W) in every Noise layer (stylegan.py:26): 13 launches of ~5 us per generator
# pass at 256 x 256.  Inside `bulk_noise(...)` the layers take consecutive slices of ONE draw instead (the same
# distribution; a different walk through the generator's random stream, so runs that replay recorded draws keep it off).
_NOISE_POOL = {"buf": None, "cursor": 0}


@contextlib.contextmanager
def bulk_noise(total: int, device, dtype):
    prev = dict(_NOISE_POOL)
    _NOISE_POOL["buf"], _NOISE_POOL["cursor"] = torch.randn(total, device=device, dtype=dtype), 0
    try:
        yield
    finally:
        _NOISE_POOL.update(prev)


def noise_draw(n: int, h: int, w: int, device, dtype) -> Tensor:
    """torch.randn(n, 1, h, w) -- or the next n*h*w values of the active bulk draw."""
    buf, cur, size = _NOISE_POOL["buf"], _NOISE_POOL["cursor"], n * h * w
    if buf is not None and buf.dtype == dtype and buf.device == device and cur + size <= buf.numel() and (cur * buf.element_size()) % 16 == 0:
        _NOISE_POOL["cursor"] = cur + size + ((-size) % 8)
        return buf[cur : cur + size].view(n, 1, h, w)
    return torch.randn(n, 1, h, w, device=device, dtype=dtype)


# ---- one memset for the accumulation targets of a whole backward pass ------------------------------------------
# gan_modconv_bwd_pre / gan_scale_dot add their workgroups' partial sums into small (N, C)-sized buffers that must
# start at zero: a memset launch per call, ~33 per generator backward.  Inside `zeroed_scratch(device)` those buffers
# are slices of one per-device pool cleared once on entry.  The pool is recycled by the next `zeroed_scratch` on the
# device, so only a caller that has consumed the pass's results by then (a trainer around loss.backward()) opts in.
_ZERO_POOL: dict = {}


def _grow_pool(st: dict, device) -> None:
    """A larger pool buffer.  The old one is retired, not freed: a captured HIP graph may have its address baked into a
    memset and into the accumulating kernels, and the caching allocator would hand freed memory to someone else while
    replays keep writing into it.  (The pools are a few hundred KB; they grow a handful of times per process.)"""
    if st["buf"] is not None:
        st.setdefault("retired", []).append(st["buf"])
    st["buf"] = torch.empty((st["want"] + st["want"] // 4 + 4096,), dtype=torch.float32, device=device)


@contextlib.contextmanager
def zeroed_scratch(device):
    device = torch.device(device)
    if device.type != "cuda":
        yield
        return
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _ZERO_POOL.setdefault(device, {"buf": None, "cursor": 0, "need": 0, "want": 0, "active": False})
    if st["active"]:
        yield
        return
    st["want"] = max(st["want"], st["need"])
    have = st["buf"].numel() if st["buf"] is not None else 0
    if st["want"] > have and not torch.cuda.is_current_stream_capturing():  # (an allocation made while capturing would belong to the graph)
        _grow_pool(st, device)
    if st["buf"] is not None:
        st["buf"].zero_()
    st["cursor"], st["need"], st["active"] = 0, 0, True
    try:
        yield
    finally:
        st["active"] = False


def grow_zero_pool() -> None:
    """Size every device's pool to what its last pass asked for (call outside stream capture)."""
    for device, st in _ZERO_POOL.items():
        st["want"] = max(st["want"], st["need"])
        if st["want"] > (st["buf"].numel() if st["buf"] is not None else 0) and not st["active"]:
            _grow_pool(st, device)


def _zeroed(nfloats: int, device) -> Optional[Tensor]:
    """A zero-filled fp32 slice of the active pool, or None (no pool active on the device, or it is exhausted: the next
    `zeroed_scratch` grows it)."""
    st = _ZERO_POOL.get(device)
    if st is None or not st["active"]:
        return None
    n = nfloats + ((-nfloats) % 64)
    st["need"] += n
    if st["buf"] is None or st["cursor"] + n > st["buf"].numel():
        return None
    out = st["buf"][st["cursor"] : st["cursor"] + nfloats]
    st["cursor"] += n
    return out


def raw_modconv_bwd_pre(gy: Tensor, y: Optional[Tensor], noise: Optional[Tensor], dinv: Optional[Tensor], slope: float):
    """One pass: gz = gy*m(y)*dinv and the per-sample sums (N, 3, C) [gy*m, gy*m*noise, gy*y] (include/gan_amd.h)."""
    require_gpu(gy, y, noise, dinv)
    lib = _lib.load()
    gy = as_cl(gy.detach())
    n, c, h, w = gy.shape
    if y is not None:
        y = as_cl(y.detach().to(gy.dtype))
    if noise is not None:
        noise = noise.detach().to(gy.dtype).reshape(n, h * w).contiguous()
    d32, _ = _side(dinv, n, c, "modconv_bwd_pre dinv")
    if d32 is not None and d32.numel() != n * c:
        d32 = d32.reshape(1, c).expand(n, c).contiguous()
    gz = _empty_like_cl(gy, gy.shape)
    zeroed = _zeroed(n * 3 * c, gy.device)
    sums = zeroed.view(n, 3, c) if zeroed is not None else torch.empty((n, 3, c), dtype=torch.float32, device=gy.device)
    check(
        _run(
            "modconv_bwd_pre", 0, _nbytes(gy, y, noise, gz),
            lambda: lib.gan_modconv_bwd_pre(ptr(gy), ptr(y), ptr(noise), ptr(d32), float(slope), ptr(gz), ptr(sums), int(zeroed is not None),
                                            n, h * w, c, dtype_code(gy), stream_of(gy)),
            (n, c, h, w),
        ),
        "gan_modconv_bwd_pre",
    )
    return gz, sums


def raw_modconv_bwd_post(sums: Tensor, bias: Optional[Tensor], ns: Optional[Tensor], dinv: Optional[Tensor],
                         want_gb: bool, want_gns: bool, want_gdinv: bool):
    """(gb (C,), gns (C,), gdinv (N, C)) in fp32 from raw_modconv_bwd_pre's sums (N, 3, C); None where not wanted."""
    require_gpu(sums, bias, ns, dinv)
    lib = _lib.load()
    n, _, c = sums.shape
    b32, s32, d32 = _f32(bias), _f32(ns), _f32(dinv)
    gb = torch.empty((c,), dtype=torch.float32, device=sums.device) if want_gb else None
    gns = torch.empty((c,), dtype=torch.float32, device=sums.device) if want_gns else None
    gdinv = torch.empty((n, c), dtype=torch.float32, device=sums.device) if want_gdinv else None
    check(lib.gan_modconv_bwd_post(ptr(sums), ptr(b32), ptr(s32), ptr(d32), ptr(gb), ptr(gns), ptr(gdinv), n, c, stream_of(sums)),
          "gan_modconv_bwd_post")
    return gb, gns, gdinv


def raw_scale_dot(A: Tensor, B: Tensor, a: Tensor):
    """One pass: out = A * a[n,c] and dot (N, C) = sum_hw A*B."""
    require_gpu(A, B, a)
    lib = _lib.load()
    A = as_cl(A.detach())
    B = as_cl(B.detach().to(A.dtype))
    n, c, h, w = A.shape
    a32 = _f32(a).reshape(n, c).contiguous()
    out = _empty_like_cl(A, A.shape)
    zeroed = _zeroed(n * c, A.device)
    dot = zeroed.view(n, c) if zeroed is not None else torch.empty((n, c), dtype=torch.float32, device=A.device)
    check(
        _run(
            "scale_dot", 0, _nbytes(A, B, out),
            lambda: lib.gan_scale_dot(ptr(A), ptr(B), ptr(a32), ptr(out), ptr(dot), int(zeroed is not None), n, h * w, c, dtype_code(A),
                                      stream_of(A)),
            (n, c, h, w),
        ),
        "gan_scale_dot",
    )
    return out, dot


# The last channel-padded copy of a tensor (RGB 3 -> 16 is a 0.5 GB pass at 256x256, batch 256): the
# forward convolution and the weight gradient of the same layer pad the same input.
_PAD_MEMO: dict = {}


def _pad_channels(x: Tensor, mult: int, dtype: Optional[torch.dtype] = None) -> Tensor:
    """Channels-last x with the channel count padded to a multiple of `mult` (zeros), in `dtype` (default x's)."""
    c = x.shape[1]
    dtype = dtype or x.dtype
    rem = (-c) % mult
    if c < 8:  # image-like input: pad once to 16 for both users (conv forward wants 8, weight gradient 16)
        key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), x.dtype, dtype)
        hit = _PAD_MEMO.get("last")
        if hit is not None and hit[0] == key:
            return hit[2]
        if x.dim() != 4:
            raise _lib.GanAmdError(f"expected a 4-D (N, C, H, W) tensor, got shape {tuple(x.shape)}")
        n, _, h, w = x.shape
        out = torch.empty((n, 16, h, w), dtype=dtype, device=x.device, memory_format=CL)
        sn, sc, sh, sw = x.stride()
        check(
            _run(
                "pad_channels", 0, _nbytes(x, out),
                lambda: _lib.load().gan_pad_channels(ptr(x), dtype_code(x), sn, sc, sh, sw, ptr(out), dtype_code(out), n, c, h, w, 16, stream_of(x)),
                (n, c, h, w),
            ),
            "gan_pad_channels",
        )
        _PAD_MEMO["last"] = (key, x, out)  # x is held: its storage cannot be recycled under the same key
        return out
    if x.dtype != dtype:
        x = x.to(dtype)
    if rem == 0:
        return as_cl(x)
    return as_cl(F.pad(x, (0, 0, 0, 0, 0, rem)))


def _packed_rows(rows: int, kpad: int, dtype: torch.dtype, device) -> Tensor:
    """Storage of a packed weight image.  Wide odd row counts (the 513 rows of the input-gradient image of the layer
    behind MinibatchStdDev, progressive_gan.py:71) get zero rows up to a multiple of 8, so that raw_conv2d can run the
    layer with a padded output-channel count on the LDS-DMA kernels instead of the register-staged fallback."""
    rows_p = rows + ((-rows) % 8) if rows > 64 else rows
    out = torch.empty((rows_p, kpad), dtype=dtype, device=device)
    if rows_p != rows:
        out[rows:].zero_()
    return out


def _pack_weight(weight: Tensor, dtype: torch.dtype, cin_mult: int, transposed: bool) -> Tuple[Tensor, int, int]:
    cout, cin, kh, kw = weight.shape
    rows, cols = (cin, cout) if transposed else (cout, cin)
    cols_p = cols + ((-cols) % cin_mult)
    k = kh * kw * cols_p
    kpad = k + ((-k) % 32)
    if weight.dtype not in (torch.float32, torch.bfloat16):
        weight = weight.float()
    weight = weight.contiguous()
    out = _packed_rows(rows, kpad, dtype, weight.device)
    check(
        _lib.load().gan_pack_weight(ptr(weight), dtype_code(weight), ptr(out), dtype_code(out), cout, cin, kh, kw, cols_p, kpad,
                                    int(transposed), stream_of(weight)),
        "gan_pack_weight",
    )
    return out, cols_p, kpad


# Packed GEMM-B images of the PARAMETERS, reused until the optimizer's in-place update bumps the tensor
# version: one conv weight is packed ~10 times per training step otherwise (forward on reals and fakes,
# both input gradients, the double-backward convolutions), ~4 tiny launches each.
_PACK_CACHE: dict = {}


def _pack_weight_pair(weight: Tensor, dtype: torch.dtype, mult_f: int, mult_t: int):
    """Both GEMM-B images of one weight in one launch: ((wt, cols_p, kpad), (wt_transposed, cols_p_t, kpad_t))."""
    cout, cin, kh, kw = weight.shape
    cp_f = cin + ((-cin) % mult_f)
    cp_t = cout + ((-cout) % mult_t)
    kf, kt = kh * kw * cp_f, kh * kw * cp_t
    kpad_f, kpad_t = kf + ((-kf) % 32), kt + ((-kt) % 32)
    weight = weight.contiguous()
    out_f = _packed_rows(cout, kpad_f, dtype, weight.device)
    out_t = _packed_rows(cin, kpad_t, dtype, weight.device)
    check(
        _lib.load().gan_pack_weight_pair(ptr(weight), dtype_code(weight), ptr(out_f), cp_f, kpad_f, ptr(out_t), cp_t, kpad_t,
                                         dtype_code(out_f), cout, cin, kh, kw, stream_of(weight)),
        "gan_pack_weight_pair",
    )
    return (out_f, cp_f, kpad_f), (out_t, cp_t, kpad_t)


# orientations (cin_mult, transposed) each parameter has been asked for: a layer that trains needs both, and
# then one launch packs both (survives _PACK_CACHE.clear(): it holds no device memory)
_PACK_SEEN: dict = {}


def pack_weight(weight: Tensor, dtype: torch.dtype, cin_mult: int = 8, transposed: bool = False) -> Tuple[Tensor, int, int]:
    """(Cout, Cin, kh, kw) -> GEMM-B layout [Cout][Kpad], k = tap*Cin_p + ci, zero padded; returns (wt, Cin_p, Kpad).
    `transposed` packs the flipped, channel-swapped weights of the input-gradient correlation."""
    cacheable = weight.is_leaf and (weight.requires_grad or isinstance(weight, torch.nn.Parameter)) and weight.numel() > 0
    if not cacheable:
        derived = _derived_owner(weight) if (_FOLD_OWNER or _SCALED_OWNER) else None
        if derived is None:
            return _pack_weight(weight.detach(), dtype, cin_mult, transposed)
        kind, owner, ver = derived
        key = (kind, id(owner), dtype, cin_mult, transposed)  # images of a parameter's folded / scaled copy: valid as long as the copy is
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0]() is owner and hit[1] == ver:
            PACK_STATS[0] += 1
            return hit[2]
        PACK_STATS[1] += 1
        packed = _pack_weight(weight.detach(), dtype, cin_mult, transposed)
        _PACK_CACHE[key] = (weakref.ref(owner), ver, packed)
        return packed
    key = (id(weight), dtype, cin_mult, transposed)
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
        PACK_STATS[0] += 1
        return hit[2]
    PACK_STATS[1] += 1
    if len(_PACK_CACHE) > 4096:
        _PACK_CACHE.clear()
    if len(_PACK_SEEN) > 4096:
        _PACK_SEEN.clear()
    seen = _PACK_SEEN.get((id(weight), dtype))
    if seen is None or seen[0]() is not weight:
        seen = _PACK_SEEN[(id(weight), dtype)] = (weakref.ref(weight), set())
    seen[1].add((cin_mult, transposed))
    other = [mt for mt in seen[1] if mt[1] != transposed]
    ref, ver = weakref.ref(weight), (weight._version, weight.data_ptr())  # data_ptr: p.data = ... does not bump the version
    if len(other) == 1 and len(seen[1]) == 2 and weight.dtype in (torch.float32, torch.bfloat16):
        mult_f, mult_t = (other[0][0], cin_mult) if transposed else (cin_mult, other[0][0])
        img_f, img_t = _pack_weight_pair(weight.detach(), dtype, mult_f, mult_t)
        _PACK_CACHE[(id(weight), dtype, mult_f, False)] = (ref, ver, img_f)
        _PACK_CACHE[(id(weight), dtype, mult_t, True)] = (ref, ver, img_t)
        return img_t if transposed else img_f
    packed = _pack_weight(weight.detach(), dtype, cin_mult, transposed)
    _PACK_CACHE[key] = (ref, ver, packed)
    return packed


PACK_STATS = [0, 0]  # hits, misses (cacheable tensors only)


def drop_packed() -> None:
    """Forget every cached GEMM image, tap sum and padded input (a graph replay or an in-place load changed the weights
    without bumping their versions; buffers recorded during an aborted capture were never written)."""
    _PACK_CACHE.clear()
    _WSQ_CACHE.clear()
    _FOLD_CACHE.clear()
    _FOLD_OWNER.clear()
    _SCALED_CACHE.clear()
    _SCALED_OWNER.clear()
    _PAD_MEMO.clear()


def invalidate_packed(params) -> None:
    """Drop the cached GEMM images of `params`: for writers that change a parameter's values behind autograd's
    back -- through a flat arena or a raw pointer (EMA.step, a graph replay) -- so that neither `_version` nor
    `data_ptr()` moves and the cache key above would still match."""
    ids = {id(p) for p in params}
    for key in [k for k in _PACK_CACHE if (k[1] if isinstance(k[0], str) else k[0]) in ids]:
        del _PACK_CACHE[key]
    for i in ids:
        _WSQ_CACHE.pop(i, None)
        hit = _FOLD_CACHE.pop(i, None)
        if hit is not None:
            _FOLD_OWNER.pop(hit[2].data_ptr(), None)
        hit = _SCALED_CACHE.pop(i, None)
        if hit is not None:
            _SCALED_OWNER.pop(hit[2][0].data_ptr(), None)


# ---- all packed images of a network in one launch -----------------------------------------------------------
# tap sums of W^2 per (co, ci) of the modulated convolutions' weights (DemodFn's first factor), by id(weight)
_WSQ_CACHE: dict = {}
_WSQ_SEEN: dict = {}
_DT_CODE = {torch.float32: _lib.GAN_F32, torch.bfloat16: _lib.GAN_BF16}
_PREPACK_GROUPS: dict = {}


def _cache_valid(key, weight: Tensor) -> bool:
    hit = _PACK_CACHE.get(key)
    return hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr())


def prepack_weights(weights) -> bool:
    """Re-pack, in ONE launch (gan_pack_weights_multi), every GEMM image the earlier passes asked of these conv weights
    (pack_weight records the orientations in _PACK_SEEN) and the tap sums DemodFn asked for, into buffers that live as
    long as the group, and point the caches at them.  A trainer calls this after an optimizer step instead of letting
    ~60 convolutions each find a stale cache entry and pack their own weight (5-8 us per launch for ~25 us of traffic).
    Returns False when nothing was launched: images all current, nothing recorded yet, or the group's table would
    have to be built (a host-to-device copy) while the stream is being captured."""
    weights = [w for w in weights if w.dim() == 4 and w.is_cuda and w.dtype in (torch.float32, torch.bfloat16) and w.is_contiguous()]
    if not weights:
        return False
    plan, stale = [], False
    for w in weights:
        wsq = _WSQ_SEEN.get(id(w))
        want_wsq = wsq is not None and wsq() is w
        if want_wsq:
            hit = _WSQ_CACHE.get(id(w))
            stale |= not (hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()))
        for dt in (torch.bfloat16, torch.float32):
            seen = _PACK_SEEN.get((id(w), dt))
            if seen is None or seen[0]() is not w:
                continue
            fwd = sorted(m for m, t in seen[1] if not t)
            tr = sorted(m for m, t in seen[1] if t)
            if len(fwd) > 1 or len(tr) > 1 or (not fwd and not tr):
                continue  # several paddings of one orientation: left to pack_weight
            if 4096 * w.shape[2] * w.shape[3] * (2 if dt == torch.bfloat16 else 4) > 150 * 1024:
                continue  # a 64 x 64 tile with all its taps must fit LDS (gan_pack_weights_multi)
            plan.append((w, dt, fwd[0] if fwd else None, tr[0] if tr else None, want_wsq))
            want_wsq = False
            for m, t in seen[1]:
                stale |= not _cache_valid((id(w), dt, m, t), w)
        if want_wsq and 4096 * w.shape[2] * w.shape[3] * 4 <= 150 * 1024:
            plan.append((w, torch.float32, None, None, True))
    if not plan or not stale:
        return False
    gkey = tuple(id(w) for w in weights)
    sig = tuple((id(w), w.data_ptr(), dt, mf, mt, q) for w, dt, mf, mt, q in plan)
    group = _PREPACK_GROUPS.get(gkey)
    if group is not None and not all(r() is w for r, w in zip(group["keep"], weights)):
        group = None  # ids recycled by other tensors
    if group is None or group["sig"] != sig:
        if torch.cuda.is_current_stream_capturing():
            return False
        for k in [k for k, g in _PREPACK_GROUPS.items() if any(r() is None for r in g["keep"])]:
            del _PREPACK_GROUPS[k]  # a group whose weights are gone: free its images
        if len(_PREPACK_GROUPS) > 64:
            _PREPACK_GROUPS.clear()
        dev = weights[0].device
        launches: dict = {}
        entries = []
        for w, dt, mf, mt, q in plan:
            cout, cin, kh, kw = w.shape
            taps = kh * kw
            img_f = img_t = wsq = None
            row = [w.data_ptr(), 0, 0, 0, 0, cout, cin, taps, 0, 0, 0, 0]
            if mf is not None:
                cp = cin + ((-cin) % mf)
                kpad = taps * cp + ((-taps * cp) % 32)
                img_f = (_packed_rows(cout, kpad, dt, dev), cp, kpad)
                row[1], row[8], row[9] = img_f[0].data_ptr(), cp, kpad
            if mt is not None:
                cp = cout + ((-cout) % mt)
                kpad = taps * cp + ((-taps * cp) % 32)
                img_t = (_packed_rows(cin, kpad, dt, dev), cp, kpad)
                row[2], row[10], row[11] = img_t[0].data_ptr(), cp, kpad
            if q:
                wsq = torch.empty((cout, cin), dtype=torch.float32, device=dev)
                row[3] = wsq.data_ptr()
            tab = launches.setdefault((w.dtype, dt), {"rows": [], "blocks": 0, "taps": 0})
            row[4] = tab["blocks"]
            tab["blocks"] += -(-max(cout, row[10]) // 64) * -(-max(cin, row[8]) // 64)  # 64 x 64 (co x ci) tiles
            tab["taps"] = max(tab["taps"], taps)
            tab["rows"].append(row)
            entries.append((weakref.ref(w), dt, mf, mt, img_f, img_t, wsq))
        for tab in launches.values():
            tab["table"] = torch.tensor(tab["rows"], dtype=torch.int64, device=dev)
        group = _PREPACK_GROUPS[gkey] = {"sig": sig, "launches": launches, "entries": entries, "keep": [weakref.ref(w) for w in weights]}
    lib = _lib.load()
    for (wdt, odt), tab in group["launches"].items():
        check(lib.gan_pack_weights_multi(ptr(tab["table"]), len(tab["rows"]), tab["blocks"], tab["taps"], _DT_CODE[wdt], _DT_CODE[odt],
                                         stream_of(tab["table"])), "gan_pack_weights_multi")
    for ref, dt, mf, mt, img_f, img_t, wsq in group["entries"]:
        w = ref()  # (alive: it is one of `weights`)
        ver = (w._version, w.data_ptr())
        if img_f is not None:
            _PACK_CACHE[(id(w), dt, mf, False)] = (ref, ver, img_f)
        if img_t is not None:
            _PACK_CACHE[(id(w), dt, mt, True)] = (ref, ver, img_t)
        if wsq is not None:
            _WSQ_CACHE[id(w)] = (ref, ver, wsq)
    if _DEMOD_WANTS and not torch.cuda.is_current_stream_capturing():
        _build_wanted_demod_tables()
    return True


def _wide_kernel_shape(x: Tensor, cin_p: int, cout: int, ho: int, wo: int, kh: int, kw: int, py: int, px: int) -> bool:
    """Mirror of the C-side dispatch (conv_fwd.hip: thin kernel first, then conv_wide.hip's eligibility)."""
    if x.dtype != torch.bfloat16 or cin_p % 64 != 0 or cout % 4 != 0 or x.shape[0] * ho * wo < 128:
        return False
    return not _thin_kernel_shape(x, cin_p, cout, ho, wo, kh, kw, py, px)


_CUS: Dict[int, int] = {}


def _halo_kernel_shape(x: Tensor, cout: int, ho: int, wo: int, kh: int, kw: int, py: int, px: int, up2: bool) -> bool:
    """Mirror of conv_wide_halo.hip's eligibility among the wide shapes: 3x3 same-size convolutions on maps of 8 / 16k
    pixels a side whose tiles fill the chip (profile family only: the library decides by itself)."""
    import os

    mode = int(os.environ.get("GAN_AMD_HALO", "1"))
    n, _, h, w = x.shape
    if mode == 0 or kh != 3 or kw != 3 or py != 1 or px != 1 or cout % 8 or ho != (2 * h if up2 else h) or wo != (2 * w if up2 else w):
        return False
    t16 = ho % 16 == 0 and wo % 16 == 0
    if not t16 and (ho % 8 or wo % 8):
        return False
    bn = 128 if cout > 64 else 64
    waves = 4 if t16 else 8
    tiles = -(-n // (1 if t16 else 4)) * (wo // (16 if t16 else 8)) * (ho // (2 * waves if t16 else 8))
    dev = x.device.index or 0
    if dev not in _CUS:
        _CUS[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    slots = _CUS[dev] * (2 if waves == 4 else 1)
    return mode == 2 or tiles * -(-cout // bn) >= slots - slots // 8


def _thin_kernel_shape(x: Tensor, cin_p: int, cout: int, ho: int, wo: int, kh: int, kw: int, py: int, px: int) -> bool:
    """Shapes conv_thin.hip serves (HBM-bound direct convolution); used to pick the profile family too."""
    return (x.dtype == torch.bfloat16 and kh == kw and kh in (1, 3) and py == px == (kh - 1) // 2
            and cin_p in (8, 16, 32, 64) and cout <= 64 and wo >= 32 and ho >= 8)


def raw_conv2d(
    x: Tensor,
    weight: Tensor,
    py: int,
    px: int,
    *,
    up2: bool = False,
    in_scale: Optional[Tensor] = None,
    out_scale: Optional[Tensor] = None,
    bias: Optional[Tensor] = None,
    noise: Optional[Tensor] = None,
    noise_scale: Optional[Tensor] = None,
    slope: float = 1.0,
    transposed: bool = False,
    in_slope: float = 1.0,
    pool: bool = False,
    aux: Optional[Tensor] = None,
    aux_slope: float = 1.0,
    gain: float = 1.0,
    resid: Optional[Tensor] = None,
) -> Tensor:
    """Stride-1 correlation of channels-last x with weight (Cout, Cin, kh, kw) + fused prologue/epilogue.
    `transposed`: correlate with the flipped, channel-swapped weights instead (weight is (Cin, Cout, kh, kw)).
    in_slope / pool / aux / gain: the fused LeakyReLU-before, AvgPool2d(2)-after, mask-after and scalar stages
    (include/gan_amd.h: gan_conv2d_fwd)."""
    require_gpu(x, weight, in_scale, out_scale, bias, noise, noise_scale)
    lib = _lib.load()
    # odd wide inputs (the 513 channels behind MinibatchStdDev) are padded to a multiple of 64 rather than 8: the
    # LDS-DMA kernel then takes the layer (110 -> ~55 us at batch 256) for 12 % more K
    mult = 64 if (x.dtype == torch.bfloat16 and x.shape[1] > 64 and x.shape[1] % 8 != 0) else 8
    x = _pad_channels(x.detach(), mult)
    n, cin_p, h, w = x.shape
    cout, cin, kh, kw = weight.shape
    if transposed:
        cout, cin = cin, cout
    wt, cin_p2, kpad = pack_weight(weight, x.dtype, cin_mult=16 if cin < 8 else mult, transposed=transposed)
    if cin_p2 != cin_p:
        raise _lib.GanAmdError(f"conv2d: input has {x.shape[1]} (padded) channels, weight expects {cin}")
    hl, wl = (2 * h, 2 * w) if up2 else (h, w)
    ho, wo = hl + 2 * py - kh + 1, wl + 2 * px - kw + 1
    if ho <= 0 or wo <= 0:
        raise _lib.GanAmdError("conv2d: empty output")
    if in_scale is not None:
        in_scale = _f32(in_scale).reshape(n, cin)
        if cin_p != cin:
            in_scale = F.pad(in_scale, (0, cin_p - cin)).contiguous()
    if (in_scale is not None or in_slope != 1.0) and _wide_kernel_shape(x, cin_p, cout, ho, wo, kh, kw, py, px):
        # The LDS-DMA kernel of the wide layers (conv_wide.hip) has no input-side stages: its operand tiles
        # never pass through registers.  Activation / modulation run as one pass over these small tensors.
        if in_slope != 1.0:
            x = raw_affine_act(x, slope=in_slope)
            in_slope = 1.0
        if in_scale is not None:
            x = raw_affine_act(x, a=in_scale)
            in_scale = None
    out_scale32, _ = _side(out_scale, n, cout, "conv2d out_scale")
    if out_scale32 is not None and out_scale32.numel() != n * cout:
        out_scale32 = out_scale32.reshape(1, cout).expand(n, cout).contiguous()
    bias32, _ = _side(bias, 1, cout, "conv2d bias")
    ns32, _ = _side(noise_scale, 1, cout, "conv2d noise scale")
    if pool and (ho % 2 or wo % 2):
        raise _lib.GanAmdError(f"conv2d: pooling needs an even output, got {ho}x{wo}")
    hy, wy = (ho // 2, wo // 2) if pool else (ho, wo)
    if noise is not None:
        noise = noise.detach().to(x.dtype).reshape(n, hy * wy).contiguous()
    if aux is not None:
        if tuple(aux.shape) != (n, cout, hy, wy):
            raise _lib.GanAmdError(f"conv2d: aux shape {tuple(aux.shape)} != output shape {(n, cout, hy, wy)}")
        aux = as_cl(aux.detach().to(x.dtype))
    if resid is not None:
        if tuple(resid.shape) != (n, cout, hy, wy):
            raise _lib.GanAmdError(f"conv2d: resid shape {tuple(resid.shape)} != output shape {(n, cout, hy, wy)}")
        resid = as_cl(resid.detach().to(x.dtype))
    # an odd, wide output-channel count (513: the input gradient of the layer behind MinibatchStdDev) with no per-channel
    # side inputs: computed with the count rounded up to 8 (the packed image has zero rows there) on the wide kernels --
    # 95 -> ~30 us per call at any batch size -- and handed back as a channel slice
    cout_real = cout
    if (x.dtype == torch.bfloat16 and cout > 64 and cout % 8 and wt.shape[0] >= cout + (-cout) % 8 and out_scale32 is None and bias32 is None
            and ns32 is None and noise is None and aux is None and resid is None):
        cout = cout + (-cout) % 8
    y = _empty_like_cl(x, (n, cout, hy, wy))
    # small-grid wide layers split K through a workspace (0 floats: this shape does not)
    ws_floats = lib.gan_conv2d_fwd_workspace(n, cin_p, ho, wo, cout, kh, kw, kpad, int(in_scale is not None), dtype_code(x))
    ws = torch.empty((ws_floats,), dtype=torch.float32, device=x.device) if ws_floats > 0 else None
    if _thin_kernel_shape(x, cin_p, cout, ho, wo, kh, kw, py, px):
        # HBM-bound; one family per device kernel instantiation conv_thin_k<CIN, COUT, KS, RPW, MODE, VEC>
        mode = "G" if (in_scale is not None or out_scale is not None or noise is not None) else ("DP" if pool else ("DA" if aux is not None else "D"))
        family = f"conv2d_fwd_thin<{cin_p},{16 if cout <= 16 else (32 if cout <= 32 else 64)},{kh},{mode}>"
    elif in_scale is None and in_slope == 1.0 and _wide_kernel_shape(x, cin_p, cout, ho, wo, kh, kw, py, px):
        # MFMA-bound: conv_wide_k<128, 2> when its 128 x 128 tiles fill the chip, else conv_wide_k<64, 2 | 4>
        # (mirror of the launch selection in conv_wide.hip)
        big = cout > 64 and ((n * ho * wo + 127) // 128) * ((cout + 127) // 128) >= 256
        family = "conv2d_fwd_wide128" if big else "conv2d_fwd_wide64"
        if _halo_kernel_shape(x, cout, ho, wo, kh, kw, py, px, up2):
            family = "conv2d_fwd_wide_halo"  # conv_wide_halo_k<BN, waves, up2>: the halo stays in LDS across the taps
    else:
        family = "conv2d_fwd"
    check(
        _run(
            family, 2.0 * n * ho * wo * cout_real * cin * kh * kw, _nbytes(x, y, wt, noise, aux, resid),
            lambda: lib.gan_conv2d_fwd_ws(
                ptr(x), ptr(wt), ptr(y), ptr(in_scale), ptr(out_scale32), ptr(bias32), ptr(noise), ptr(ns32), float(slope),
                n, h, w, cin_p, int(up2), ho, wo, cout, kh, kw, py, px, kpad,
                float(in_slope), int(pool), ptr(aux), float(aux_slope), float(gain), ptr(resid), ptr(ws), ws_floats, dtype_code(x), stream_of(x),
            ),
            (n, cin, ho, wo, cout_real, kh, int(up2), int(pool)),
        ),
        "gan_conv2d_fwd",
    )
    return y if cout == cout_real else y[:, :cout_real]


_FUSE_SHORTCUT = os.environ.get("GAN_AMD_FUSE_SC", "1") != "0"  # A/B runs: 0 = two kernels + residual read


def raw_dblock_input_grad(g_p1: Tensor, w1: Tensor, pad: int, x_aux: Tensor, slope: float, g_s: Tensor, ws: Tensor):
    """The input gradient of a D block's two consumers of x (progressive_gan.py:45-52) in one kernel (gan_conv2d_fwd_short):
    m(x_aux) * dgrad3x3(g_p1, w1) + 1/4 NearestUp2(dgrad1x1(g_s, ws)) -- or None where the kernel does not serve the shape.
    w1: (C, C, 3, 3), ws: (2C, C, 1, 1), g_p1 (N, C, H, W), g_s (N, 2C, H/2, W/2)."""
    if not _FUSE_SHORTCUT or g_p1.dtype != torch.bfloat16 or not g_p1.is_cuda:
        return None
    lib = _lib.load()
    c1o, c, kh, kw = w1.shape
    sc_c = ws.shape[0]
    n, cz, h, w = g_p1.shape
    if cz != c1o or cz % 8 or tuple(ws.shape[1:]) != (c, 1, 1) or tuple(g_s.shape) != (n, sc_c, h // 2, w // 2) or tuple(x_aux.shape) != (n, c, h, w):
        return None
    kpad_guess = (kh * kw * cz + 31) // 32 * 32
    if lib.gan_conv2d_fwd_short_supported(n, h, w, cz, c, kh, kw, kh - 1 - pad, kw - 1 - pad, kpad_guess, sc_c, dtype_code(g_p1)) != 1:
        return None
    gz = as_cl(g_p1.detach())
    gs = as_cl(g_s.detach().to(gz.dtype))
    wt, cz_p, kpad = pack_weight(w1, gz.dtype, cin_mult=8, transposed=True)
    swt, sc_p, skpad = pack_weight(ws, gz.dtype, cin_mult=8, transposed=True)
    if cz_p != cz or kpad != kpad_guess or sc_p != sc_c or skpad != sc_c:
        return None
    aux = as_cl(x_aux.detach().to(gz.dtype))
    y = _empty_like_cl(gz, (n, c, h, w))
    family = f"conv2d_fwd_thin<{cz},{16 if c <= 16 else (32 if c <= 32 else 64)},{kh},DAS>"
    check(
        _run(
            family, 2.0 * n * h * w * c * (cz * kh * kw + sc_c // 4), _nbytes(gz, y, wt, aux, gs),
            lambda: lib.gan_conv2d_fwd_short(ptr(gz), ptr(wt), ptr(y), ptr(aux), float(slope), 1.0, ptr(gs), ptr(swt), sc_c, 0.25, n, h, w, cz, c,
                                             kh, kw, kh - 1 - pad, kw - 1 - pad, kpad, dtype_code(gz), stream_of(gz)),
            (n, cz, h, w, c, kh, 0, 0),
        ),
        "gan_conv2d_fwd_short",
    )
    return y


_FUSE_SCALE_DOT = os.environ.get("GAN_AMD_FUSE_SD", "1") != "0"  # A/B runs: 0 = raw_conv2d + raw_scale_dot


def raw_conv2d_dgrad_scale_dot(gz: Tensor, weight: Tensor, py: int, px: int, s: Tensor, xin: Tensor, *, pool: bool = False, gain: float = 1.0):
    """The input gradient of a modulated convolution with the backward of its `x * s` on the accumulators (gan_conv2d_fwd_dot):
    gxs = conv(gz, weight^T flipped, pad k-1-p) * gain [2x2-pooled], returns (gxs * s[n, c], sum_hw gxs * xin) -- or None where the
    kernel does not serve the shape (the caller then runs raw_conv2d + raw_scale_dot).  weight: (Cout, Cin, kh, kw) of the layer."""
    require_gpu(gz, weight, s, xin)
    lib = _lib.load()
    if gz.dtype != torch.bfloat16 or not _FUSE_SCALE_DOT:
        return None
    gz = _pad_channels(gz.detach(), 8)
    n, cz_p, h, w = gz.shape
    cz, cin, kh, kw = weight.shape  # the layer's (Cout, Cin): this convolution maps Cout -> Cin
    if cin % 4 or cz_p > 64 or tuple(xin.shape[:2]) != (n, cin):
        return None
    ho, wo = h + 2 * py - kh + 1, w + 2 * px - kw + 1
    hy, wy = (ho // 2, wo // 2) if pool else (ho, wo)
    if tuple(xin.shape[2:]) != (hy, wy):
        return None
    kpad_guess = (kh * kw * cz_p + 31) // 32 * 32
    if lib.gan_conv2d_fwd_dot_supported(n, h, w, cz_p, ho, wo, cin, kh, kw, py, px, kpad_guess, int(pool), dtype_code(gz)) != 1:
        return None
    wt, cz_p2, kpad = pack_weight(weight, gz.dtype, cin_mult=16 if cz < 8 else 8, transposed=True)
    if cz_p2 != cz_p or kpad != kpad_guess:
        return None
    s32 = _f32(s).reshape(n, cin).contiguous()
    xc = as_cl(xin.detach().to(gz.dtype))
    y = _empty_like_cl(gz, (n, cin, hy, wy))
    zeroed = _zeroed(n * cin, gz.device)
    dot = zeroed.view(n, cin) if zeroed is not None else torch.zeros((n, cin), dtype=torch.float32, device=gz.device)
    family = f"conv2d_fwd_thin<{cz_p},{16 if cin <= 16 else (32 if cin <= 32 else 64)},{kh},{'DPS' if pool else 'DS'}>"
    check(
        _run(
            family, 2.0 * n * ho * wo * cin * cz * kh * kw, _nbytes(gz, y, wt, xc),
            lambda: lib.gan_conv2d_fwd_dot(ptr(gz), ptr(wt), ptr(y), ptr(s32), ptr(xc), ptr(dot), n, h, w, cz_p, ho, wo, cin, kh, kw, py, px, kpad,
                                           int(pool), float(gain), dtype_code(gz), stream_of(gz)),
            (n, cz, ho, wo, cin, kh, 0, int(pool)),
        ),
        "gan_conv2d_fwd_dot",
    )
    return y, dot


def raw_conv2d_dgrad(
    gy: Tensor, weight: Tensor, py: int, px: int, *, out_scale: Optional[Tensor] = None,
    pooled: bool = False, aux: Optional[Tensor] = None, aux_slope: float = 1.0, resid: Optional[Tensor] = None,
) -> Tensor:
    """Input gradient of a stride-1 correlation = correlation of gy with the transposed, flipped
    weights and padding k-1-p.  Output spatial size is the forward input size.
    pooled: gy is the gradient of the 2x2-average-pooled output (nearest-2x gather, gain 1/4);
    aux: multiply by the leaky-ReLU mask of aux (the activation that preceded the convolution)."""
    cout, cin, kh, kw = weight.shape
    return raw_conv2d(gy, weight, kh - 1 - py, kw - 1 - px, out_scale=out_scale, transposed=True,
                      up2=pooled, gain=0.25 if pooled else 1.0, aux=aux, aux_slope=aux_slope, resid=resid)


def raw_conv2d_wgrad(
    x: Tensor, gy: Tensor, kh: int, kw: int, py: int, px: int, *, up2: bool = False, in_scale: Optional[Tensor] = None,
    x_slope: float = 1.0, pooled: bool = False, into: Optional[Tensor] = None, bias: bool = False, bias_into: Optional[Tensor] = None,
):
    """fp32 (Cout, Cin, kh, kw).  x_slope: x goes through leaky-ReLU first; pooled: gy is the gradient of the
    2x2-average-pooled output (half resolution); into: an fp32 (Cout, Cin, kh, kw) buffer the gradient is ADDED to.
    bias: also the layer's bias gradient, sum of gy over (n, y, x), from the same call (gan_conv2d_wgrad_bias; Cout a
    multiple of 16) -> returns (dw, dbias); bias_into: an fp32 (Cout,) buffer it is ADDED to."""
    require_gpu(x, gy, in_scale)
    lib = _lib.load()
    cin, cout = x.shape[1], gy.shape[1]
    gy = gy.detach().to(x.dtype)
    x = _pad_channels(x.detach(), 16)
    gy = _pad_channels(gy, 16)
    n, cin_p, h, w = x.shape
    cout_p, ho, wo = gy.shape[1], gy.shape[2], gy.shape[3]
    if pooled:
        ho, wo = 2 * ho, 2 * wo
    if in_scale is not None:
        in_scale = _f32(in_scale).reshape(n, cin)
        if cin_p != cin:
            in_scale = F.pad(in_scale, (0, cin_p - cin)).contiguous()
    if into is not None:
        assert into.dtype == torch.float32 and into.is_contiguous() and tuple(into.shape) == (cout, cin, kh, kw)
    dw = into if into is not None else torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=x.device)
    thin_w = (x.dtype == torch.bfloat16 and kh == kw and kh in (1, 3) and py == px == (kh - 1) // 2 and cin_p <= 64
              and cout_p <= 64 and wo >= 32 and ho >= 8)
    if bias:
        if cout_p != cout:
            raise _lib.GanAmdError(f"conv2d_wgrad: the fused bias gradient needs Cout % 16 == 0, got {cout}")
        if bias_into is not None:
            assert bias_into.dtype == torch.float32 and bias_into.is_contiguous() and bias_into.numel() == cout
        db = bias_into if bias_into is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)
        ws_floats = lib.gan_conv2d_wgrad_bias_workspace(n, h, w, cin_p, int(up2), ho, wo, cout_p, kh, kw, py, px, int(pooled), dtype_code(x))
        ws = torch.empty((max(ws_floats, 1),), dtype=torch.float32, device=x.device)
        check(
            _run(
                "conv2d_wgrad_thin" if thin_w else "conv2d_wgrad", 2.0 * n * ho * wo * cout * cin * kh * kw, _nbytes(x, gy, dw),
                lambda: lib.gan_conv2d_wgrad_bias(
                    ptr(x), ptr(gy), ptr(dw), cout, cin, ptr(db), int(bias_into is not None), ptr(in_scale), n, h, w, cin_p, int(up2), ho, wo,
                    cout_p, kh, kw, py, px, ptr(ws), ws_floats, float(x_slope), int(pooled), 0.25 if pooled else 1.0, int(into is not None),
                    dtype_code(x), stream_of(x)
                ),
                (n, cin, ho, wo, cout, kh, int(up2), int(pooled)),
            ),
            "gan_conv2d_wgrad_bias",
        )
        return dw, db
    # split-K partials go to a caller-owned workspace (the library never allocates); torch's caching
    # allocator hands the same block back every step.  The slice reducer writes dw in the parameter's layout.
    ws_floats = lib.gan_conv2d_wgrad_workspace(n, h, w, cin_p, int(up2), ho, wo, cout_p, kh, kw, py, px, dtype_code(x))
    ws = torch.empty((ws_floats,), dtype=torch.float32, device=x.device) if ws_floats > 0 else None
    check(
        _run(
            "conv2d_wgrad_thin" if thin_w else "conv2d_wgrad", 2.0 * n * ho * wo * cout * cin * kh * kw, _nbytes(x, gy, dw),
            lambda: lib.gan_conv2d_wgrad(
                ptr(x), ptr(gy), ptr(dw), cout, cin, ptr(in_scale), n, h, w, cin_p, int(up2), ho, wo, cout_p, kh, kw, py, px,
                ptr(ws), ws_floats, float(x_slope), int(pooled), 0.25 if pooled else 1.0, int(into is not None), dtype_code(x), stream_of(x)
            ),
            (n, cin, ho, wo, cout, kh, int(up2), int(pooled)),
        ),
        "gan_conv2d_wgrad",
    )
    return dw


def raw_mbstd_stat(x: Tensor, group: int) -> Tensor:
    require_gpu(x)
    lib = _lib.load()
    x = as_cl(x.detach())
    n, c, h, w = x.shape
    stat = torch.empty((n // group,), dtype=torch.float32, device=x.device)
    check(lib.gan_mbstd_fwd(ptr(x), ptr(stat), n, group, h * w, c, dtype_code(x), stream_of(x)), "gan_mbstd_fwd")
    return stat


def raw_ema_lerp(ema_flat: Tensor, p_flat: Tensor, weight: float) -> None:
    require_gpu(ema_flat, p_flat)
    lib = _lib.load()
    assert ema_flat.dtype == torch.float32 and p_flat.dtype == torch.float32
    assert ema_flat.is_contiguous() and p_flat.is_contiguous() and ema_flat.numel() == p_flat.numel()
    check(lib.gan_ema_lerp(ptr(ema_flat), ptr(p_flat), ema_flat.numel(), float(weight), stream_of(ema_flat)), "gan_ema_lerp")


# ======================================================================================
# differentiable primitives
# ======================================================================================
class Function(_TorchFunction):
    """torch.autograd.Function whose node remembers which positional arguments were tensors: the node's gradient edges
    (`next_functions`) are numbered over the tensor arguments only, `needs_input_grad` over all of them (see _need)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        out = super().apply(*args, **kwargs)
        first = out if isinstance(out, Tensor) else next((o for o in out if isinstance(o, Tensor)), None) if isinstance(out, (tuple, list)) else None
        node = getattr(first, "grad_fn", None)
        if node is not None and isinstance(node, cls._backward_cls):
            pos, k = [], 0
            for a in args:
                if isinstance(a, Tensor):
                    pos.append(k)
                    k += 1
                else:
                    pos.append(-1)
            node._gan_edge_of_arg = tuple(pos)
        return out


def _need(ctx, i: int) -> bool:
    """Does THIS backward pass want the gradient of input i?  `ctx.needs_input_grad` is fixed at forward time (does the input
    require grad at all): under `torch.autograd.grad(out, inputs=[reals], create_graph=True)` -- the first pass of R1 and of
    the WGAN gradient penalty (losses.py:95-103) -- it still says True for every weight and bias, and a backward that believes
    it computes a weight gradient and a bias sum per layer that the engine then throws away (a third of the discriminator's
    weight-gradient kernels per iteration).  The engine knows which nodes the running graph task will execute: an input whose
    gradient edge leads to a node it will not execute is not wanted."""
    if not ctx.needs_input_grad[i]:
        return False
    if not _PRUNE_UNWANTED:
        return True
    edge = getattr(ctx, "_gan_edge_of_arg", None)
    if edge is None or i >= len(edge) or edge[i] < 0:
        return True
    try:
        fn = ctx.next_functions[edge[i]][0]
        return True if fn is None else bool(torch._C._will_engine_execute_node(fn))
    except Exception:  # (leaf nodes under autograd.grad raise: those are asked for by name; anything unexpected: compute)
        return True


_PRUNE_UNWANTED = os.environ.get("GAN_AMD_PRUNE_GRADS", "1") != "0" and hasattr(torch._C, "_will_engine_execute_node")  # A/B runs: 0


# ---- parameter gradients added into their buffers by the kernels that compute them -------------------------
# autograd sums the contributions a parameter receives in one backward pass with one `add` launch per extra
# contribution and tensor (a discriminator weight gets three per step: the real batch, the fake batch, R1's
# second-order path): ~75 weight-sized launches per iteration.  Inside `direct_param_grads()`, a backward pass that
# is not itself recorded (no create_graph) hands conv weights / biases whose `.grad` is already an fp32 buffer to
# the split-K fold / channel-sum kernels with accumulate = 1 and returns None for them: same sums, no launches.
# Parameter hooks do not see these gradients (post-accumulate hooks never fire), so the hook-driven bucketed
# all-reduce path leaves this off; the trainer's graph path (flat all-reduce after the pass) turns it on.
_DIRECT_PARAM_GRADS = False


@contextlib.contextmanager
def direct_param_grads(enabled: bool = True):
    global _DIRECT_PARAM_GRADS
    prev, _DIRECT_PARAM_GRADS = _DIRECT_PARAM_GRADS, enabled
    try:
        yield
    finally:
        _DIRECT_PARAM_GRADS = prev


def grad_slot(p: Optional[Tensor]) -> Optional[Tensor]:
    """p.grad when the current backward pass may add p's gradient straight into it, else None."""
    if not _DIRECT_PARAM_GRADS or p is None or torch.is_grad_enabled() or not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not g.is_cuda:
        return None
    return g


def _wgrad_for(w: Tensor, x: Tensor, gy: Tensor, py: int, px: int, x_slope: float = 1.0, pool: bool = False) -> Optional[Tensor]:
    """The gradient of conv weight w (Conv2dWgradFn), or None after adding it into w.grad (grad_slot)."""
    slot = grad_slot(w)
    if slot is not None:
        raw_conv2d_wgrad(x, gy, w.shape[2], w.shape[3], py, px, x_slope=x_slope, pooled=pool, into=slot)
        return None
    return Conv2dWgradFn.apply(x, gy, w.shape[2], w.shape[3], py, px, x_slope, pool).to(w.dtype)


def _bias_grad_for(b: Tensor, g: Tensor) -> Optional[Tensor]:
    """The gradient of a per-channel bias (ChanSumFn over the batch), or None after adding it into b.grad."""
    slot = grad_slot(b)
    if slot is not None and slot.numel() == g.shape[1]:
        raw_chan_dot(g, None, 0, None, 1.0, False, into=slot)
        return None
    if isinstance(b, torch.nn.Parameter):
        b._gan_direct = True  # FlatArena.begin_direct: give this parameter a zeroed gradient slice from the next pass on
    return ChanSumFn.apply(g, False).reshape(b.shape).to(b.dtype)


def _wgrad_and_bias_for(w: Tensor, b: Tensor, x: Tensor, gy: Tensor, py: int, px: int, x_slope: float = 1.0, pool: bool = False):
    """(gradient of conv weight w, gradient of its bias b) from the same gy.  In a backward pass that is not recorded both
    come from ONE call (gan_conv2d_wgrad_bias: on thin layers the kernel sums gy with one more MFMA per tile row, so the
    channel-sum pass over gy -- a full read of the largest tensors of the discriminator's backward -- disappears); each is
    None when it was added into the parameter's .grad (grad_slot)."""
    if torch.is_grad_enabled() or gy.shape[1] % 16 != 0 or b.numel() != gy.shape[1] or not gy.is_cuda:
        return _wgrad_for(w, x, gy, py, px, x_slope, pool), _bias_grad_for(b, gy)
    wslot, bslot = grad_slot(w), grad_slot(b)
    if isinstance(b, torch.nn.Parameter):
        b._gan_direct = True
    dw, db = raw_conv2d_wgrad(x, gy, w.shape[2], w.shape[3], py, px, x_slope=x_slope, pooled=pool, into=wslot, bias=True, bias_into=bslot)
    return (None if wslot is not None else dw.to(w.dtype)), (None if bslot is not None else db.reshape(b.shape).to(b.dtype))


class UpFirDnFn(Function):
    """upfirdn2d with a general, shape-exact adjoint.  The reference's UpFIRDn2d.backward
    (nvidia_ops.py:50-54) calls itself with the flipped kernel, up<->down and swapped pads, which is
    the true adjoint only when p1 + p2 == k - 1; here the backward pads are derived from the shapes
    (begin pad k-1-p1, output size = forward input size), which equals the reference wherever the
    reference's backward runs and is also right elsewhere.  No gradient for the FIR taps
    (nvidia_ops.py:54)."""

    @staticmethod
    def forward(ctx, x, taps, up, down, px1, px2, py1, py2, ho=None, wo=None):
        n, c, h, w = x.shape
        ky, kx = taps.shape
        if ho is None:
            ho = (h * up + py1 + py2 - ky) // down + 1
            wo = (w * up + px1 + px2 - kx) // down + 1
        if ho <= 0 or wo <= 0:
            raise _lib.GanAmdError(f"upfirdn2d: empty output for input {h}x{w}")
        ctx.save_for_backward(taps)
        ctx.cfg = (up, down, px1, py1, h, w, ho, wo, ky, kx)
        return raw_upfirdn2d(x, taps, up, down, px1, py1, ho, wo)

    @staticmethod
    def backward(ctx, gy):
        (taps,) = ctx.saved_tensors
        up, down, px1, py1, h, w, ho, wo, ky, kx = ctx.cfg
        gx = None
        if _need(ctx, 0):
            a1x, a1y = kx - 1 - px1, ky - 1 - py1
            a2x = (w - 1) * up + kx - wo * down - a1x
            a2y = (h - 1) * up + ky - ho * down - a1y
            flipped = taps if taps.numel() == 4 and taps.data_ptr() in _CONST_TAPS else taps.flip(0, 1)  # constant taps are symmetric
            gx = UpFirDnFn.apply(gy, flipped, down, up, a1x, a2x, a1y, a2y, h, w)
        return (gx,) + (None,) * 9


class MaskMulFn(Function):
    """g * m(y), m(y) = 1 where y > 0 else slope: the leaky-ReLU backward as a function of the
    activation OUTPUT.  Linear in g; the mask has zero derivative almost everywhere."""

    @staticmethod
    def forward(ctx, g, y, slope):
        ctx.save_for_backward(y)
        ctx.slope = slope
        return raw_mask_scale(g, y, None, slope)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return (MaskMulFn.apply(gg, y, ctx.slope) if _need(ctx, 0) else None), None, None


class LreluFn(Function):
    """nn.LeakyReLU(slope), out of place (stylegan2.py:49,56; progressive_gan.py:44,46,73)."""

    @staticmethod
    def forward(ctx, x, slope):
        y = raw_affine_act(x, slope=slope)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return MaskMulFn.apply(gy, y, ctx.slope), None


class ChanScaleFn(Function):
    """x * a, a of shape (C,) or (N, C) broadcast over pixels.  Bilinear."""

    @staticmethod
    def forward(ctx, x, a):
        ctx.save_for_backward(x, a)
        return raw_mask_scale(x, None, a, 1.0)

    @staticmethod
    def backward(ctx, gy):
        x, a = ctx.saved_tensors
        gx = ChanScaleFn.apply(gy, a) if _need(ctx, 0) else None
        ga = None
        if _need(ctx, 1):
            ga = ChanDotFn.apply(gy, x, a.numel() != x.shape[1]).reshape(a.shape).to(a.dtype)
        return gx, ga


class ChanDotFn(Function):
    """sum over pixels (and over the batch unless per_sample) of A*B -> fp32 (N, C) or (C,).  Bilinear."""

    @staticmethod
    def forward(ctx, A, B, per_sample):
        ctx.save_for_backward(A, B)
        ctx.per_sample = per_sample
        return raw_chan_dot(A, B, 1, None, 1.0, per_sample)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        gA = ChanScaleFn.apply(B, g).to(A.dtype) if _need(ctx, 0) else None
        gB = ChanScaleFn.apply(A, g).to(B.dtype) if _need(ctx, 1) else None
        return gA, gB, None


class ChanSumFn(Function):
    """sum over pixels (and batch unless per_sample) -> fp32 (N, C) or (C,): bias / per-channel gradients."""

    @staticmethod
    def forward(ctx, A, per_sample):
        ctx.shape = A.shape
        ctx.dtype = A.dtype
        ctx.per_sample = per_sample
        return raw_chan_dot(A, None, 0, None, 1.0, per_sample)

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = g.reshape(n if ctx.per_sample else 1, c, 1, 1).to(ctx.dtype)
        return g.expand(n, c, h, w).contiguous(memory_format=CL), None


class PixelDotFn(Function):
    """sum_{n,hw} A[n,c,hw] * noise[n,hw] -> fp32 (C,): gradient of Noise.scale (stylegan.py:22,27).
    The noise draw is a constant."""

    @staticmethod
    def forward(ctx, A, noise):
        ctx.save_for_backward(noise)
        return raw_chan_dot(A, noise, 2, None, 1.0, False)

    @staticmethod
    def backward(ctx, g):
        (noise,) = ctx.saved_tensors
        # d/dA = noise[n,hw] * g[c]
        n, _, h, w = noise.shape
        gA = noise.to(torch.float32) * g.reshape(1, -1, 1, 1)
        return gA.to(noise.dtype).contiguous(memory_format=CL), None


class Conv2dFn(Function):
    """y = AvgPool2?( conv( leaky_relu?(x), W ) ): stride-1 correlation with zero padding (py, px), no bias
    (biases ride in the fused epilogues).  The activation BEFORE the convolution (in_slope != 1) and the 2x2
    average pooling AFTER it are fused into the kernel: the discriminator block's LeakyReLU -> Conv2d and
    LeakyReLU -> Conv2d -> AvgPool2d chains (progressive_gan.py:45-50) are one launch each.
    Reference arithmetic: F.conv2d as used by nn.Conv2d and by batched_conv2d (stylegan2.py:18,36).

    Conv2dFn / Conv2dDgradFn / Conv2dWgradFn are closed under differentiation: for fixed activation masks
    the map is bilinear in (x, W), and leaky-ReLU has no second derivative."""

    @staticmethod
    def forward(ctx, x, weight, py, px, in_slope=1.0, pool=False):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (py, px, in_slope, pool)
        return raw_conv2d(x, weight, py, px, in_slope=in_slope, pool=pool)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        py, px, in_slope, pool = ctx.cfg
        xaux = x if in_slope != 1.0 else None
        gx = Conv2dDgradFn.apply(gy, weight, py, px, xaux, in_slope, pool) if _need(ctx, 0) else None
        gw = None
        if _need(ctx, 1):
            gw = _wgrad_for(weight, x, gy, py, px, in_slope, pool)
        return gx, gw, None, None, None, None


class Conv2dDgradFn(Function):
    """gx = m(xaux) * dgrad( PoolAdj?(gy), W ) [+ resid]: the adjoint of Conv2dFn in x.  PoolAdj (nearest-2x
    gather, gain 1/4), the mask of the activation that preceded the convolution and the gradient arriving at
    the same tensor along another path (`resid`) are fused into the kernel."""

    @staticmethod
    def forward(ctx, gy, weight, py, px, xaux=None, slope=1.0, pool=False, resid=None):
        ctx.save_for_backward(gy, weight, xaux)
        ctx.cfg = (py, px, slope, pool)
        return raw_conv2d_dgrad(gy, weight, py, px, pooled=pool, aux=xaux, aux_slope=slope, resid=resid)

    @staticmethod
    def backward(ctx, ggx):
        gy, weight, xaux = ctx.saved_tensors
        py, px, slope, pool = ctx.cfg
        z = MaskMulFn.apply(ggx, xaux, slope) if xaux is not None else ggx
        g_gy = Conv2dFn.apply(z, weight, py, px, 1.0, pool).to(gy.dtype) if _need(ctx, 0) else None
        g_w = None
        if _need(ctx, 1):
            g_w = _wgrad_for(weight, z, gy, py, px, 1.0, pool)
        g_resid = ggx if len(ctx.needs_input_grad) > 7 and ctx.needs_input_grad[7] else None
        return g_gy, g_w, None, None, None, None, None, g_resid


class DBlockInputGradFn(Function):
    """gx = m(x) * dgrad3x3(g_p1, W1) + 1/4 NearestUp2(dgrad1x1(g_s, Ws)): the input gradient of a thin D block's two consumers
    of x (DBlockInputFn below) as ONE kernel (gan_conv2d_fwd_short) -- the shortcut's gradient is never written out.  Bilinear in
    (g_p1, W1) and (g_s, Ws) for a fixed mask; its backward is the pair of Conv2dDgradFn backwards it replaces, so R1's second
    order runs through it too."""

    @staticmethod
    def forward(ctx, g_p1, g_s, w1, ws, x, pad, slope):
        ctx.save_for_backward(g_p1, g_s, w1, ws, x)
        ctx.cfg = (pad, slope)
        y = raw_dblock_input_grad(g_p1, w1, pad, x, slope, g_s, ws)
        if y is None:  # (fp32, odd shapes: the two kernels, the shortcut's gradient as the residual of the other)
            gx_s = raw_conv2d_dgrad(g_s, ws, 0, 0, pooled=True)
            y = raw_conv2d_dgrad(g_p1, w1, pad, pad, aux=x, aux_slope=slope, resid=gx_s)
        return y

    @staticmethod
    def backward(ctx, ggx):
        g_p1, g_s, w1, ws, x = ctx.saved_tensors
        pad, slope = ctx.cfg
        z = MaskMulFn.apply(ggx, x, slope)
        g_gp1 = Conv2dFn.apply(z, w1, pad, pad, 1.0, False).to(g_p1.dtype) if _need(ctx, 0) else None
        g_gs = Conv2dFn.apply(ggx, ws, 0, 0, 1.0, True).to(g_s.dtype) if _need(ctx, 1) else None
        g_w1 = _wgrad_for(w1, z, g_p1, pad, pad, 1.0, False) if _need(ctx, 2) else None
        g_ws = _wgrad_for(ws, ggx, g_s, 0, 0, 1.0, True) if _need(ctx, 3) else None
        return g_gp1, g_gs, g_w1, g_ws, None, None, None


class DBlockInputFn(Function):
    """The two consumers of a discriminator block's input x (progressive_gan.py:45-46,52):
         p1 = conv3x3(leaky_relu(x), W1) + b1          s = AvgPool2d(2) -> conv1x1(Ws)
    as ONE autograd node, so that their two gradients with respect to x are produced by one kernel: the
    shortcut's input gradient rides as the residual of conv1's input-gradient convolution instead of autograd
    adding two full-resolution tensors.  Backward is composed of the Functions above (differentiable again)."""

    @staticmethod
    def forward(ctx, x, w1, b1, ws, slope, pool_in_conv, bs=None):
        pad = (w1.shape[2] - 1) // 2
        p1 = raw_conv2d(x, w1, pad, pad, bias=b1, in_slope=slope)
        if pool_in_conv:  # thin layers: conv1x1 -> pool in one launch (the same linear map)
            xs = None
            s = raw_conv2d(x, ws, 0, 0, pool=True, bias=bs)
        else:             # wide layers: the 1x1 GEMM on a quarter of the pixels
            xs = raw_upfirdn2d(x, _ones_taps(x, 0.25), 1, 2, 0, 0, x.shape[2] // 2, x.shape[3] // 2)
            s = raw_conv2d(xs, ws, 0, 0, bias=bs)
        ctx.save_for_backward(x, w1, b1, ws, xs)
        ctx.cfg = (pad, slope, pool_in_conv)
        ctx.bs = bs  # (a parameter: held by the module anyway)
        return p1, s

    @staticmethod
    def backward(ctx, g_p1, g_s):
        x, w1, b1, ws, xs = ctx.saved_tensors
        pad, slope, pool_in_conv = ctx.cfg
        gx = gw1 = gb1 = gws = gbs = None
        if _need(ctx, 0):
            if pool_in_conv or (w1.shape[0] == 64 and w1.shape[1] == 64 and x.dtype == torch.bfloat16 and x.shape[3] >= 32):
                # thin conv1 (64 -> 64 too, whose shortcut ran on the wide kernels forward): both input gradients out of one kernel
                gx = DBlockInputGradFn.apply(g_p1, g_s, w1, ws, x, pad, slope)
            else:
                gx_s = UpFirDnFn.apply(Conv2dDgradFn.apply(g_s, ws, 0, 0), _ones_taps(x, 0.25), 2, 1, 1, 0, 1, 0)
                gx = Conv2dDgradFn.apply(g_p1, w1, pad, pad, x, slope, False, gx_s)
        if _need(ctx, 1) and b1 is not None and _need(ctx, 2):
            gw1, gb1 = _wgrad_and_bias_for(w1, b1, x, g_p1, pad, pad, slope, False)
        else:
            if _need(ctx, 1):
                gw1 = _wgrad_for(w1, x, g_p1, pad, pad, slope, False)
            if b1 is not None and _need(ctx, 2):
                gb1 = _bias_grad_for(b1, g_p1)
        want_bs = ctx.bs is not None and _need(ctx, 6)
        if _need(ctx, 3):
            xs_in, pooled = (x, True) if pool_in_conv else (xs, False)
            if want_bs:
                gws, gbs = _wgrad_and_bias_for(ws, ctx.bs, xs_in, g_s, 0, 0, 1.0, pooled)
            else:
                gws = _wgrad_for(ws, xs_in, g_s, 0, 0, 1.0, pooled)
        elif want_bs:
            gbs = _bias_grad_for(ctx.bs, g_s)
        return gx, gw1, gb1, gws, None, None, gbs


_FUSE_DBLOCK = os.environ.get("GAN_AMD_FUSE_DBLOCK", "1") != "0"  # A/B runs: 0 = DBlockInputFn + ConvBiasActFn


def _shortcut_fwd(x: Tensor, ws: Tensor, bs: Optional[Tensor], pool_in_conv: bool):
    """AvgPool2d(2) -> conv1x1 (progressive_gan.py:52); thin layers: conv1x1 -> pool in one launch (the same linear map).
    Returns (s, pooled x or None)."""
    if pool_in_conv:
        return raw_conv2d(x, ws, 0, 0, pool=True, bias=bs), None
    xs = raw_upfirdn2d(x, _ones_taps(x, 0.25), 1, 2, 0, 0, x.shape[2] // 2, x.shape[3] // 2)
    return raw_conv2d(xs, ws, 0, 0, bias=bs), xs


class DBlockGradFn(Function):
    """(gx, g_p1) of a whole discriminator block for the gradient gy of its output (progressive_gan.py:43-54, scale folded
    into conv2: DBlockFn below):
         g_p1 = m(p1) * dgrad3x3(PoolAdj(gy), W2s)        gx = m(x) * dgrad3x3(g_p1, W1) + PoolAdj(dgrad1x1(gy, Ws))
    Forward: the kernels Conv2dDgradFn / DBlockInputGradFn run.  Backward -- R1's second-order pass, bilinear in (gy, W) for
    fixed masks -- as ONE chain instead of the composition of those nodes' backwards:
         z1 = m(x) * ggx;  z2 = m(p1) * conv(z1, W1) [the mask on the convolution's accumulators];
         g_gy = Pool(conv(z2, W2s)) + conv1x1(Pool(ggx), Ws) [the shortcut term as the convolution's residual]
    i.e. no mask pass between the two convolutions and no sum of two gradient tensors by the autograd engine (two
    activation-sized passes per block and R1 step).  Weight gradients: wgrad(z1, g_p1), wgrad(Pool ggx, gy), wgrad(z2, gy)."""

    @staticmethod
    def forward(ctx, gy, x, p1, w1, ws, w2s, pad, slope0, slope1, pool_in_conv):
        ctx.set_materialize_grads(False)
        g_p1 = raw_conv2d_dgrad(gy, w2s, pad, pad, pooled=True, aux=p1, aux_slope=slope1)
        gx = None
        if pool_in_conv or (w1.shape[0] == 64 and w1.shape[1] == 64 and x.dtype == torch.bfloat16 and x.shape[3] >= 32):
            gx = raw_dblock_input_grad(g_p1, w1, pad, x, slope0, gy, ws)
        if gx is None:
            if pool_in_conv:
                gx_s = raw_conv2d_dgrad(gy, ws, 0, 0, pooled=True)
            else:
                gx_s = raw_upfirdn2d(raw_conv2d_dgrad(gy, ws, 0, 0), _ones_taps(x, 0.25), 2, 1, 1, 1, x.shape[2], x.shape[3])
            gx = raw_conv2d_dgrad(g_p1, w1, pad, pad, aux=x, aux_slope=slope0, resid=gx_s)
        ctx.save_for_backward(gy, x, p1, g_p1, w1, ws, w2s)
        ctx.cfg = (pad, slope0, slope1, pool_in_conv)
        return gx, g_p1

    @staticmethod
    def backward(ctx, ggx, gg_p1):
        gy, x, p1, g_p1, w1, ws, w2s = ctx.saved_tensors
        pad, slope0, slope1, pool_in_conv = ctx.cfg
        need_gy, need_w1, need_ws, need_w2 = _need(ctx, 0), _need(ctx, 3), _need(ctx, 4), _need(ctx, 5)
        if torch.is_grad_enabled():
            # a third derivative is being recorded: the differentiable pieces (never in the train step)
            g_gy = g_w1 = g_ws = g_w2 = None
            u = gg_p1
            if ggx is not None:
                z1 = MaskMulFn.apply(ggx, x, slope0)
                c1 = Conv2dFn.apply(z1, w1, pad, pad, 1.0, False)
                u = c1 if u is None else c1 + u
                g_w1 = Conv2dWgradFn.apply(z1, g_p1, 3, 3, pad, pad, 1.0, False).to(w1.dtype) if need_w1 else None
                g_gy = Conv2dFn.apply(ggx, ws, 0, 0, 1.0, True).to(gy.dtype)
                g_ws = Conv2dWgradFn.apply(ggx, gy, 1, 1, 0, 0, 1.0, True).to(ws.dtype) if need_ws else None
            if u is not None:
                z2 = MaskMulFn.apply(u, p1, slope1)
                t = Conv2dFn.apply(z2, w2s, pad, pad, 1.0, True).to(gy.dtype)
                g_gy = t if g_gy is None else g_gy + t
                g_w2 = Conv2dWgradFn.apply(z2, gy, 3, 3, pad, pad, 1.0, True).to(w2s.dtype) if need_w2 else None
            return g_gy, None, None, g_w1, g_ws, g_w2, None, None, None, None
        z1 = raw_mask_scale(ggx, x, None, slope0) if ggx is not None else None
        z2 = None
        if need_gy or need_w2:
            if z1 is not None and gg_p1 is None:
                z2 = raw_conv2d(z1, w1, pad, pad, aux=p1, aux_slope=slope1)
            elif z1 is not None:
                z2 = raw_mask_scale(raw_conv2d(z1, w1, pad, pad) + gg_p1.to(z1.dtype), p1, None, slope1)
            elif gg_p1 is not None:
                z2 = raw_mask_scale(gg_p1, p1, None, slope1)
        g_gy = None
        if need_gy:
            gs2 = _shortcut_fwd(ggx, ws, None, pool_in_conv)[0] if ggx is not None else None
            g_gy = raw_conv2d(z2, w2s, pad, pad, pool=True, resid=gs2) if z2 is not None else gs2
            if g_gy is not None:
                g_gy = g_gy.to(gy.dtype)
        g_w1 = _wgrad_for(w1, z1, g_p1, pad, pad, 1.0, False) if need_w1 and z1 is not None else None
        g_ws = _wgrad_for(ws, ggx, gy, 0, 0, 1.0, True) if need_ws and ggx is not None else None
        g_w2 = _wgrad_for(w2s, z2, gy, pad, pad, 1.0, True) if need_w2 and z2 is not None else None
        return g_gy, None, None, g_w1, g_ws, g_w2, None, None, None, None


class DBlockFn(Function):
    """A discriminator block (progressive_gan.py:43-54) as ONE autograd node:
         p1 = lrelu(conv3x3(lrelu(x), W1) + b1);  s = conv1x1(AvgPool(x), Ws) + bs;  out = AvgPool(conv3x3(p1, W2s)) + b2s + s
    with (W2s, b2s) = scaled_conv_params(W2, b2, scale).  Three launches forward, as DBlockInputFn + ConvBiasActFn; the point
    is the backward: the block's input gradient is DBlockGradFn, whose own backward (R1's second order) is one chain."""

    @staticmethod
    def forward(ctx, x, w1, b1, ws, bs, w2s, b2s, slope0, slope1, pool_in_conv):
        pad = (w1.shape[2] - 1) // 2
        # p1 is stored ACTIVATED (conv1's output stage applies the second LeakyReLU): conv2 is its only consumer, the masks
        # m(p1) read signs only, and the wide kernels -- which have no input-side stage -- need no activation pass in front of
        # conv2 (12 activation-sized launches per iteration)
        p1 = raw_conv2d(x, w1, pad, pad, bias=b1, in_slope=slope0, slope=slope1)
        s, xs = _shortcut_fwd(x, ws, bs, pool_in_conv)
        out = raw_conv2d(p1, w2s, pad, pad, bias=b2s, pool=True, resid=s)
        ctx.save_for_backward(x, p1, xs, w1, b1, ws, bs, w2s, b2s)
        ctx.cfg = (pad, slope0, slope1, pool_in_conv)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, p1, xs, w1, b1, ws, bs, w2s, b2s = ctx.saved_tensors
        pad, slope0, slope1, pool_in_conv = ctx.cfg

        def params(w, b, iw, ib, xin, g, py, x_slope, pooled):
            want_b = b is not None and _need(ctx, ib)
            if _need(ctx, iw) and want_b:
                return _wgrad_and_bias_for(w, b, xin, g, py, py, x_slope, pooled)
            return (_wgrad_for(w, xin, g, py, py, x_slope, pooled) if _need(ctx, iw) else None), (_bias_grad_for(b, g) if want_b else None)

        gw2, gb2 = params(w2s, b2s, 5, 6, p1, gy, pad, 1.0, True)
        xs_in, pooled = (x, True) if pool_in_conv else (xs, False)
        gws, gbs = params(ws, bs, 3, 4, xs_in, gy, 0, 1.0, pooled)
        gx = g_p1 = None
        need_c1 = _need(ctx, 1) or (b1 is not None and _need(ctx, 2))
        if _need(ctx, 0):
            gx, g_p1 = DBlockGradFn.apply(gy, x, p1, w1, ws, w2s, pad, slope0, slope1, pool_in_conv)
        elif need_c1:
            g_p1 = Conv2dDgradFn.apply(gy, w2s, pad, pad, p1, slope1, True)
        gw1, gb1 = params(w1, b1, 1, 2, x, g_p1, pad, slope0, False) if need_c1 else (None, None)
        return gx, gw1, gb1, gws, gbs, gw2, gb2, None, None, None


class Conv2dWgradFn(Function):
    """gW = wgrad( leaky_relu?(x), PoolAdj?(gy) ) in fp32: the adjoint of Conv2dFn in W."""

    @staticmethod
    def forward(ctx, x, gy, kh, kw, py, px, x_slope=1.0, pool=False):
        ctx.save_for_backward(x, gy)
        ctx.cfg = (kh, kw, py, px, x_slope, pool)
        return raw_conv2d_wgrad(x, gy, kh, kw, py, px, x_slope=x_slope, pooled=pool)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        kh, kw, py, px, x_slope, pool = ctx.cfg
        xaux = x if x_slope != 1.0 else None
        g_x = Conv2dDgradFn.apply(gy, ggw, py, px, xaux, x_slope, pool).to(x.dtype) if _need(ctx, 0) else None
        g_gy = Conv2dFn.apply(x, ggw, py, px, x_slope, pool).to(gy.dtype) if _need(ctx, 1) else None
        return g_x, g_gy, None, None, None, None, None, None


class AffineActFn(Function):
    """Fused y = lrelu_slope(x*a + b + noise*ns + r) in one HBM pass (kernel gan_affine_act_fwd).
    a: (C,) or (N, C); b, ns: (C,); noise: (N, 1, H, W) constant; r: residual like x.
    Backward is composed of the primitives above, so it can be differentiated again."""

    @staticmethod
    def forward(ctx, x, a, b, noise, ns, r, slope):
        y = raw_affine_act(x, a, b, noise, ns, r, slope)
        ctx.save_for_backward(x, a, b, noise, ns, r, y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        x, a, b, noise, ns, r, y = ctx.saved_tensors
        gz = MaskMulFn.apply(gy, y, ctx.slope) if ctx.slope != 1.0 else gy
        gx = ga = gb = gns = gr = None
        if _need(ctx, 0):
            gx = ChanScaleFn.apply(gz, a) if a is not None else gz
        if a is not None and _need(ctx, 1):
            ga = ChanDotFn.apply(gz, x, a.numel() != x.shape[1]).reshape(a.shape).to(a.dtype)
        if b is not None and _need(ctx, 2):
            gb = ChanSumFn.apply(gz, False).reshape(b.shape).to(b.dtype)
        if ns is not None and _need(ctx, 4):
            gns = PixelDotFn.apply(gz, noise).reshape(ns.shape).to(ns.dtype)
        if r is not None and _need(ctx, 5):
            gr = gz.to(r.dtype)
        return gx, ga, gb, None, gns, gr, None


# ======================================================================================
# the generator's mapping network and style split (gan_amd/csrc/mapping.hip)
# ======================================================================================
def _ptr_array(tensors):
    import ctypes

    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class MappingFn(Function):
    """w = mapping_network(z): LayerNorm(128) then L x [Linear(128, 128), LeakyReLU(slope)] (stylegan2.py:70-72,93) in one
    launch; backward in two.  Inputs: z, ln weight, ln bias, then W_0, b_0, ..., W_{L-1}, b_{L-1}.  No gradient for z
    (callers whose z needs one, and backward passes that are themselves recorded, take the library path instead)."""

    @staticmethod
    def forward(ctx, z, ln_w, ln_b, slope, eps, *wb):
        lib = _lib.load()
        ws, bs = list(wb[0::2]), list(wb[1::2])
        n, dim = z.shape
        L = len(ws)
        z = z.detach().contiguous()
        acts = torch.empty((L + 1, n, dim), dtype=torch.float32, device=z.device)
        stats = torch.empty((n, 2), dtype=torch.float32, device=z.device)
        check(lib.gan_mapping_fwd(ptr(z), ptr(ln_w), ptr(ln_b), _ptr_array(ws), _ptr_array(bs), L, ptr(acts), ptr(stats), n, dim, float(slope),
                                  float(eps), stream_of(z)), "gan_mapping_fwd")
        ctx.save_for_backward(z, ln_w, acts, stats, *ws)
        ctx.slope, ctx.has_bias = slope, [b is not None for b in bs]
        return acts[L]

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():
            raise _lib.GanAmdError("MappingFn: this backward pass is being recorded (create_graph); use the module path (mapping_network(...))")
        lib = _lib.load()
        z, ln_w, acts, stats, *ws = ctx.saved_tensors
        n, dim = z.shape
        L = len(ws)
        g = g.detach().to(torch.float32).contiguous()
        grads = torch.empty((L * (dim * dim + dim) + 2 * dim,), dtype=torch.float32, device=z.device)
        wsp_floats = lib.gan_mapping_bwd_workspace(n, L)
        wsp = torch.empty((wsp_floats,), dtype=torch.float32, device=z.device)
        check(lib.gan_mapping_bwd(ptr(g), ptr(acts), ptr(stats), ptr(z), ptr(ln_w), _ptr_array(ws), L, ptr(grads), ptr(wsp), wsp_floats, n, dim,
                                  float(ctx.slope), stream_of(z)), "gan_mapping_bwd")
        dW = grads[: L * dim * dim].view(L, dim, dim)
        dB = grads[L * dim * dim : L * (dim * dim + dim)].view(L, dim)
        tail = grads[L * (dim * dim + dim) :]
        out = [None, tail[:dim] if ctx.needs_input_grad[1] else None, tail[dim:] if ctx.needs_input_grad[2] else None, None, None]
        for l in range(L):
            out.append(dW[l] if ctx.needs_input_grad[5 + 2 * l] else None)
            out.append(dB[l] if ctx.has_bias[l] and ctx.needs_input_grad[6 + 2 * l] else None)
        return tuple(out)


def mapping_network(net: torch.nn.Sequential, z: Tensor) -> Tensor:
    """`net(z)` for the reference's mapping network; the fused kernels when it is LayerNorm(128) + up to 8 x [Linear(128, 128),
    LeakyReLU] in fp32 on the GPU, z needs no gradient and the pass is not under create_graph -- else the modules as they are."""
    mods = list(net)
    ok = (len(mods) >= 3 and len(mods) % 2 == 1 and isinstance(mods[0], torch.nn.LayerNorm) and z.is_cuda and z.dim() == 2 and z.shape[1] == 128
          and tuple(mods[0].normalized_shape) == (128,) and not z.requires_grad and (len(mods) - 1) // 2 <= 8 and z.shape[0] > 0)
    lins, slope = [], None
    if ok:
        for lin, act in zip(mods[1::2], mods[2::2]):
            if not (isinstance(lin, torch.nn.Linear) and isinstance(act, torch.nn.LeakyReLU) and lin.in_features == 128 and lin.out_features == 128
                    and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous() and act.negative_slope > 0
                    and (slope is None or slope == act.negative_slope)):
                ok = False
                break
            slope = act.negative_slope
            lins.append(lin)
    if not ok:
        return net(z)
    ln = mods[0]
    wb = []
    for lin in lins:
        wb += [lin.weight, lin.bias]
    return MappingFn.apply(z.float(), ln.weight, ln.bias, slope, ln.eps, *wb)


_SPLIT_OFFSETS: dict = {}


def _split_offsets(device, sizes):
    """(device int32 tensor, list) of the column offsets of SplitColsFn for these sizes; uploaded once per size tuple."""
    key = (device, tuple(sizes))
    off = _SPLIT_OFFSETS.get(key)
    if off is None:
        acc = [0]
        for c in sizes:
            acc.append(acc[-1] + c)
        off = _SPLIT_OFFSETS[key] = (torch.tensor(acc, dtype=torch.int32, device=device), acc)
    return off


class SplitColsFn(Function):
    """(N, T) fp32 -> one contiguous (N, C_l) tensor per layer (views of one buffer), one launch; backward: one torch.cat."""

    @staticmethod
    def forward(ctx, x, sizes):
        require_gpu(x)
        lib = _lib.load()
        n, t = x.shape
        off = _split_offsets(x.device, sizes)
        x = x.detach().to(torch.float32).contiguous()
        out = torch.empty((n * t,), dtype=torch.float32, device=x.device)
        check(lib.gan_split_cols(ptr(x), ptr(out), ptr(off[0]), len(sizes), n, t, 0, stream_of(x)), "gan_split_cols")
        ctx.sizes = tuple(sizes)
        return tuple(out[n * a : n * b].view(n, b - a) for a, b in zip(off[1][:-1], off[1][1:]))

    @staticmethod
    def backward(ctx, *grads):
        n = next(g.shape[0] for g in grads if g is not None)
        dev = next(g.device for g in grads if g is not None)
        full = [g if g is not None else torch.zeros((n, c), dtype=torch.float32, device=dev) for g, c in zip(grads, ctx.sizes)]
        return torch.cat(full, dim=1), None


# ======================================================================================
# functional API
# ======================================================================================
def upfirdn2d(imgs: Tensor, kernel: Tensor, up: int, down: int, px1: int, px2: int, py1: int, py2: int) -> Tensor:
    return UpFirDnFn.apply(imgs, kernel, up, down, px1, px2, py1, py2)


_TAPS: dict = {}
_CONST_TAPS: set = set()


def _ones_taps(x: Tensor, value: float) -> Tensor:
    """Constant 2x2 taps, one tensor per (device, value) for the life of the process (they were a fill
    launch per nearest-up / avg-pool call)."""
    key = (x.device, value)
    t = _TAPS.get(key)
    if t is None:
        t = _TAPS[key] = torch.full((2, 2), value, dtype=torch.float32, device=x.device)
        _CONST_TAPS.add(t.data_ptr())
    return t


def nearest_up2(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2.0) as upfirdn2d(ones(2,2), up=2, pads (1,0,1,0))."""
    return UpFirDnFn.apply(x, _ones_taps(x, 1.0), 2, 1, 1, 0, 1, 0)


class Up2AddFn(Function):
    """nearest_up2(lo) + hi in one pass (the generator's RGB skip, stylegan2.py:99).  Backward: hi's gradient is g
    itself, lo's is the 2x2 block sum of g (UpFirDnFn, so it can be differentiated again)."""

    @staticmethod
    def forward(ctx, lo, hi):
        require_gpu(lo, hi)
        lo_c = as_cl(lo.detach())
        hi_c = as_cl(hi.detach().to(lo_c.dtype))
        n, c, h, w = lo_c.shape
        if tuple(hi_c.shape) != (n, c, 2 * h, 2 * w):
            raise _lib.GanAmdError(f"up2_add: hi shape {tuple(hi_c.shape)} is not twice lo's {tuple(lo_c.shape)}")
        out = torch.empty_like(hi_c)
        check(
            _run("up2_add", 0, _nbytes(lo_c, hi_c, out),
                 lambda: _lib.load().gan_up2_add(ptr(lo_c), ptr(hi_c), ptr(out), n, h, w, c, dtype_code(lo_c), stream_of(lo_c)), (n, c, h, w)),
            "gan_up2_add",
        )
        ctx.hi_dtype = hi.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g_lo = UpFirDnFn.apply(g, _ones_taps(g, 1.0), 1, 2, 0, 0, 0, 0) if _need(ctx, 0) else None
        g_hi = g.to(ctx.hi_dtype) if _need(ctx, 1) else None
        return g_lo, g_hi


def up2_add(lo: Tensor, hi: Tensor) -> Tensor:
    return Up2AddFn.apply(lo, hi)


def avg_pool2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(2) as upfirdn2d(ones(2,2)/4, down=2, no pad)."""
    return UpFirDnFn.apply(x, _ones_taps(x, 0.25), 1, 2, 0, 0, 0, 0)


def leaky_relu(x: Tensor, slope: float = 0.2) -> Tensor:
    return LreluFn.apply(x, slope)


def bias_leaky_relu(x: Tensor, bias: Optional[Tensor], slope: float = 0.2) -> Tensor:
    """Fused lrelu(x + bias[c]): the op north_star names; oracle = F.leaky_relu(x + b.view(1,-1,1,1), 0.2)."""
    return AffineActFn.apply(x, None, bias, None, None, None, slope)


def affine_act(x, a=None, b=None, noise=None, ns=None, r=None, slope: float = 1.0) -> Tensor:
    return AffineActFn.apply(x, a, b, noise, ns, r, slope)


def conv2d(x: Tensor, weight: Tensor, padding: int = 0) -> Tensor:
    return Conv2dFn.apply(x, weight, padding, padding)


def act_dtype(x: Tensor) -> torch.dtype:
    """Compute dtype of the kernels: the autocast dtype when autocast is on (train_celeba.py:92), else x's."""
    if torch.is_autocast_enabled("cuda") if hasattr(torch, "is_autocast_enabled") else False:
        return torch.get_autocast_dtype("cuda")
    return x.dtype


class ConvBiasActFn(Function):
    """lrelu_slope(conv(lrelu_in(x), W) + b): the activation before the convolution applied as the input is
    staged, bias and the activation after it applied to the MFMA accumulators (one launch, one store).
    Replaces nn.LeakyReLU -> nn.Conv2d [-> nn.LeakyReLU] (progressive_gan.py:45-48,72-73).
    Backward = mask_mul -> dgrad / wgrad / channel sum, all differentiable again."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, slope, in_slope=1.0, pool=False, resid=None):
        if resid is not None and slope != 1.0:
            raise _lib.GanAmdError("conv_bias_act: a residual rides behind a linear output stage only (slope = 1)")
        y = raw_conv2d(x, weight, pad, pad, bias=bias, slope=slope, in_slope=in_slope, pool=pool, resid=resid)
        ctx.save_for_backward(x, weight, bias, y if slope != 1.0 else None)
        ctx.cfg = (pad, slope, in_slope, pool)
        ctx.resid_dtype = resid.dtype if resid is not None else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, y = ctx.saved_tensors
        pad, slope, in_slope, pool = ctx.cfg
        gz = MaskMulFn.apply(gy, y, slope) if slope != 1.0 else gy
        xaux = x if in_slope != 1.0 else None
        gx = Conv2dDgradFn.apply(gz, weight, pad, pad, xaux, in_slope, pool) if _need(ctx, 0) else None
        gw = gb = None
        if _need(ctx, 1) and bias is not None and _need(ctx, 2):
            gw, gb = _wgrad_and_bias_for(weight, bias, x, gz, pad, pad, in_slope, pool)
        else:
            if _need(ctx, 1):
                gw = _wgrad_for(weight, x, gz, pad, pad, in_slope, pool)
            if bias is not None and _need(ctx, 2):
                gb = _bias_grad_for(bias, gz)
        g_resid = gy.to(ctx.resid_dtype) if ctx.resid_dtype is not None and _need(ctx, 7) else None  # (added last: the identity)
        return gx, gw, gb, None, None, None, None, g_resid


def conv_bias_act(x: Tensor, weight: Tensor, bias: Optional[Tensor], padding: int, slope: float = 1.0, in_slope: float = 1.0,
                  pool: bool = False, resid: Optional[Tensor] = None) -> Tensor:
    """pool: AvgPool2d(2) of the convolution result, BEFORE the bias and the activation (avg_pool(z + b) = avg_pool(z) + b).
    resid: a tensor shaped like the result, added last (slope = 1 only): a residual merge without a pass of its own."""
    return ConvBiasActFn.apply(x, weight, bias, padding, slope, in_slope, pool, resid)


# ---- conv parameters scaled per output channel: the D block's `shortcut + layers * scale` (progressive_gan.py:54) -------
# conv(x, W) * scale + b * scale = conv(x, W * scale) + b * scale: with the scale folded into the block's last convolution
# and the shortcut added as that convolution's residual, the merge and its adjoints (a scale pass over the incoming
# gradient, a channel dot product with the saved branch output -- every D pass, first and second order) cost nothing
# activation-sized; what remains is weight-sized: one launch forward per version of the parameters, one backward.
_SCALED_CACHE: dict = {}   # id(w) -> (weakref(w), version key, (ws, bs), version function)
_SCALED_OWNER: dict = {}   # data_ptr of a cached ws -> weakref(w): lets pack_weight cache the GEMM images of ws


class ScaleConvParamsFn(Function):
    """(W (Cout, Cin, kh, kw), b (Cout,) | None, s (Cout, ...)) -> (W * s[co], b * s[co]), fp32 (kernel gan_scale_rows).
    Backward (kernel gan_scale_rows_bwd): gW = gWs * s, gb = gbs * s, gs = rowdot(gWs, W) + gbs * b; added straight into
    the parameters' gradient buffers under direct_param_grads().  Under create_graph: the same formulas on tensor ops
    (weight-sized; no caller in the train step -- R1 / WGAN-GP differentiate with respect to images).
    The node keeps the parameters on ctx (no saved-tensor slots): one cached output serves every forward pass of a
    step (reals, fakes) and survives a backward pass."""

    @staticmethod
    def forward(ctx, w, b, s):
        require_gpu(w, b, s)
        lib = _lib.load()
        if w.dtype != torch.float32 or s.dtype != torch.float32 or (b is not None and b.dtype != torch.float32) or s.numel() != w.shape[0]:
            raise _lib.GanAmdError("scaled_conv_params: fp32 parameters, one scale per output channel")
        wd, sd = w.detach().contiguous(), s.detach().reshape(-1).contiguous()
        bd = b.detach().contiguous() if b is not None else None
        ws = torch.empty_like(wd)
        bs = torch.empty_like(bd) if bd is not None else None
        check(lib.gan_scale_rows(ptr(wd), ptr(bd), ptr(sd), ptr(ws), ptr(bs), w.shape[0], wd.numel() // max(w.shape[0], 1), stream_of(wd)),
              "gan_scale_rows")
        ctx.params = (w, b, s)
        ctx.set_materialize_grads(False)
        if bs is None:
            return ws, None
        return ws, bs

    @staticmethod
    def backward(ctx, gws, gbs):
        w, b, s = ctx.params
        if gws is None and gbs is None:
            return None, None, None
        need_w, need_b, need_s = _need(ctx, 0), b is not None and _need(ctx, 1), _need(ctx, 2)
        if torch.is_grad_enabled():
            sv = s.reshape(-1)
            gw = gws * sv.view(-1, 1, 1, 1) if need_w and gws is not None else None
            gb = gbs * sv if need_b and gbs is not None else None
            gs = None
            if need_s:
                gs = torch.zeros_like(sv)
                if gws is not None:
                    gs = gs + (gws * w).flatten(1).sum(1)
                if gbs is not None and b is not None:
                    gs = gs + gbs * b
                gs = gs.reshape(s.shape)
            return gw, gb, gs
        lib = _lib.load()
        rows, cols = w.shape[0], w.numel() // max(w.shape[0], 1)
        wd, sd = w.detach().contiguous(), s.detach().reshape(-1).contiguous()
        bd = b.detach().contiguous() if b is not None else None
        g_w = gws.detach().float().contiguous() if gws is not None else None
        g_b = gbs.detach().float().contiguous() if (gbs is not None and b is not None) else None
        wslot = grad_slot(w) if need_w and g_w is not None else None
        bslot = grad_slot(b) if need_b and g_b is not None else None
        sslot = grad_slot(s) if need_s else None
        for p_ in (b, s):  # FlatArena.begin_direct: zeroed gradient slices for these from the next pass on
            if isinstance(p_, torch.nn.Parameter):
                p_._gan_direct = True
        gw = wslot if wslot is not None else (torch.empty_like(wd) if need_w and g_w is not None else None)
        gb = bslot if bslot is not None else (torch.empty_like(bd) if need_b and g_b is not None else None)
        gs = sslot.reshape(-1) if sslot is not None else (torch.empty_like(sd) if need_s else None)
        check(lib.gan_scale_rows_bwd(ptr(g_w), ptr(g_b), ptr(wd), ptr(bd), ptr(sd), ptr(gw), int(wslot is not None), ptr(gb), int(bslot is not None),
                                     ptr(gs), int(sslot is not None), rows, cols, stream_of(wd)), "gan_scale_rows_bwd")
        return (None if wslot is not None else gw), (None if bslot is not None else gb), (None if sslot is not None or gs is None else gs.reshape(s.shape))


def scaled_conv_params(w: Tensor, b: Optional[Tensor], s: Tensor):
    """(w * s[co], b * s[co]) for a conv weight / bias and a per-output-channel scale, computed once per version of the
    three (cached on the weight: the reals and fakes passes of a discriminator step share one copy and its packed GEMM
    images; an optimizer step, a graph replay or a checkpoint load drops it through invalidate_packed / drop_packed)."""
    needs = torch.is_grad_enabled() and (w.requires_grad or s.requires_grad or (b is not None and b.requires_grad))
    if not (isinstance(w, torch.nn.Parameter) and w.is_leaf and s.is_leaf and (b is None or b.is_leaf)):
        ws, bs = ScaleConvParamsFn.apply(w, b, s)
        return ws, bs
    wr, sr, br = weakref.ref(w), weakref.ref(s), (weakref.ref(b) if b is not None else None)

    def version():
        w_, s_, b_ = wr(), sr(), (br() if br is not None else None)
        if w_ is None or s_ is None or (br is not None and b_ is None):
            return None
        return (w_._version, w_.data_ptr(), s_._version, s_.data_ptr(), (b_._version, b_.data_ptr()) if b_ is not None else None, needs)

    ver = version()
    hit = _SCALED_CACHE.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == ver:
        return hit[2]
    if hit is not None:
        _SCALED_OWNER.pop(hit[2][0].data_ptr(), None)
    if len(_SCALED_CACHE) > 1024:
        _SCALED_CACHE.clear()
        _SCALED_OWNER.clear()
    out = ScaleConvParamsFn.apply(w, b, s)
    _SCALED_CACHE[id(w)] = (wr, ver, out, version)
    _SCALED_OWNER[out[0].data_ptr()] = wr
    return out


class DemodFn(Function):
    """dinv (N, Cout) = 1 / max(||W * s[n]||_(ci,kh,kw), 1e-8): the demodulation factor of ModulatedConv2d
    (stylegan2.py:33) from s (N, Cin) and W (Cout, Cin, kh, kw), fp32, two launches forward and two backward
    (kernels gan_demod_fwd / gan_demod_bwd).  Under create_graph the backward is the same formula written with
    tensor ops on the inputs, so it can be differentiated again."""

    @staticmethod
    def forward(ctx, s, weight):
        require_gpu(s, weight)
        lib = _lib.load()
        s32 = s.detach().float().contiguous()
        w32 = weight.detach().float().contiguous()
        n, cin = s32.shape
        cout, taps = w32.shape[0], w32.shape[2] * w32.shape[3]
        dinv = torch.empty((n, cout), dtype=torch.float32, device=s.device)
        hit = _WSQ_CACHE.get(id(weight))
        if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr()):
            wsq = hit[2]  # written with the packed images (prepack_weights)
            check(lib.gan_demod_fwd(ptr(s32), None, ptr(wsq), ptr(dinv), n, cin, cout, taps, stream_of(s32)), "gan_demod_fwd")
        else:
            if isinstance(weight, torch.nn.Parameter) and weight.dtype == torch.float32:
                if len(_WSQ_SEEN) > 4096:
                    _WSQ_SEEN.clear()
                _WSQ_SEEN[id(weight)] = weakref.ref(weight)
            wsq = torch.empty((cout, cin), dtype=torch.float32, device=s.device)
            check(lib.gan_demod_fwd(ptr(s32), ptr(w32), ptr(wsq), ptr(dinv), n, cin, cout, taps, stream_of(s32)), "gan_demod_fwd")
        ctx.save_for_backward(s, weight, wsq, dinv)
        return dinv

    @staticmethod
    def backward(ctx, g):
        s, weight, wsq, dinv = ctx.saved_tensors
        if torch.is_grad_enabled():
            sf, wf = s.float(), weight.float()
            wsq_t = wf.square().sum(dim=(2, 3))                                   # (Cout, Cin)
            d = (sf.square() @ wsq_t.t()).rsqrt()                                  # (N, Cout)
            dq = torch.where(d < 1e8, -0.5 * d * d * d * g.float(), torch.zeros_like(d))
            gs = (2.0 * sf * (dq @ wsq_t)).to(s.dtype) if _need(ctx, 0) else None
            gw = (2.0 * wf * (dq.t() @ sf.square())[:, :, None, None]).to(weight.dtype) if _need(ctx, 1) else None
            return gs, gw
        lib = _lib.load()
        s32 = s.detach().float().contiguous()
        w32 = weight.detach().float().contiguous()
        n, cin = s32.shape
        cout, taps = w32.shape[0], w32.shape[2] * w32.shape[3]
        g = g.detach().float().contiguous()
        gs = torch.empty_like(s32) if _need(ctx, 0) else None
        slot = grad_slot(weight) if _need(ctx, 1) and weight.dtype == torch.float32 else None
        gw = slot if slot is not None else (torch.empty_like(w32) if _need(ctx, 1) else None)
        check(lib.gan_demod_bwd(ptr(g), ptr(dinv), ptr(s32), ptr(w32), ptr(wsq), ptr(gs), ptr(gw), int(slot is not None), n, cin, cout, taps,
                                stream_of(g)), "gan_demod_bwd")
        return (gs.to(s.dtype) if gs is not None else None), (gw.to(weight.dtype) if gw is not None and slot is None else None)


# ---- every demodulating layer of a generator at once (gan_demod_multi) --------------------------------------------
_DEMOD_TABLES: dict = {}
_DEMOD_WANTS: dict = {}


def _build_wanted_demod_tables() -> None:
    for key, (refs, soffs, n) in list(_DEMOD_WANTS.items()):
        ws = [r() for r in refs]
        if any(w is None for w in ws) or _demod_tables(ws, soffs, n) is not None:
            del _DEMOD_WANTS[key]
_FUSE_DEMOD = os.environ.get("GAN_AMD_FUSE_DEMOD", "1") != "0"  # A/B runs: 0 = one DemodFn per layer


def _demod_tables(weights, soffs, n: int):
    """The three launch tables (factor, style gradient, weight gradient) of gan_demod_multi for these weights at batch n, or
    None when they cannot be had: a weight without current tap sums from prepack_weights, or tables that would have to be
    uploaded while the stream is being captured.  The weight-gradient table points at the parameters' CURRENT .grad buffers
    (the flat arena's slices: stable); the backward uses it only if the pass may add in place and they are still those."""
    wsqs, slots = [], []
    for w in weights:
        hit = _WSQ_CACHE.get(id(w))
        if hit is None or hit[0]() is not w or hit[1] != (w._version, w.data_ptr()) or w.dtype != torch.float32 or not w.is_contiguous():
            return None
        wsqs.append(hit[2])
        g = w.grad if isinstance(w, torch.nn.Parameter) else None
        slots.append(g if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == w.shape and g.is_cuda else None)
    key = (n, tuple(soffs), tuple(w.data_ptr() for w in weights), tuple(q.data_ptr() for q in wsqs),
           tuple(g.data_ptr() if g is not None else 0 for g in slots))
    hit = _DEMOD_TABLES.get(key)
    if hit is not None:
        return hit
    if torch.cuda.is_current_stream_capturing():
        return None
    if len(_DEMOD_TABLES) > 64:
        _DEMOD_TABLES.clear()
    dev = weights[0].device
    doffs, acc = [], 0
    for w in weights:
        doffs.append(acc)
        acc += w.shape[0]
    tabs = []
    for mode in range(3):
        rows, blocks = [], 0
        for w, q, g, so, do in zip(weights, wsqs, slots, soffs, doffs):
            cout, cin, kh, kw = w.shape
            m_, n_ = ((n, cout), (n, cin), (cout, cin))[mode]
            rows.append([q.data_ptr(), w.data_ptr(), g.data_ptr() if g is not None else 0, cin, cout, kh * kw, so, do, blocks, 1])
            blocks += -(-n_ // 16) * -(-m_ // 16)
        tabs.append((torch.tensor(rows, dtype=torch.int64, device=dev), blocks))
    _split_offsets(dev, tuple(w.shape[0] for w in weights))  # (demod_all splits the factors: uploaded here, outside any capture)
    out = _DEMOD_TABLES[key] = {"tabs": tabs, "doffs": doffs, "ldd": acc, "wsqs": list(wsqs),
                                "slot_ptrs": [g.data_ptr() if g is not None else 0 for g in slots]}
    return out


class DemodAllFn(Function):
    """dinv_all (N, sum Cout_l) = the demodulation factors (DemodFn) of every demodulating layer l of a generator, from the
    matrix s_all (N, sum Cin over all modulated layers) that holds every layer's style (stylegan2.py:30: one GEMM, all layers
    see the same w) -- layer l's style starts at column soffs[l].  One launch forward, two backward (gan_demod_multi) instead
    of one / two per layer: these are (N, C)-sized products whose cost is their launch, the same at any batch.  Weight
    gradients go straight into the parameters' buffers (direct_param_grads); under create_graph the backward is DemodFn's
    formula on column slices."""

    @staticmethod
    def forward(ctx, s_all, soffs, tables, *weights):
        lib = _lib.load()
        s32 = s_all.detach().float().contiguous()
        n, lds = s32.shape
        dinv_all = torch.empty((n, tables["ldd"]), dtype=torch.float32, device=s32.device)
        tab, blocks = tables["tabs"][0]
        check(lib.gan_demod_multi(0, ptr(tab), len(weights), blocks, ptr(s32), lds, None, None, tables["ldd"], ptr(dinv_all), n, stream_of(s32)),
              "gan_demod_multi")
        ctx.save_for_backward(s_all, dinv_all, *weights)
        ctx.soffs, ctx.tables = tuple(soffs), tables
        return dinv_all

    @staticmethod
    def backward(ctx, g_all):
        s_all, dinv_all, *weights = ctx.saved_tensors
        soffs, tables = ctx.soffs, ctx.tables
        doffs = tables["doffs"]
        n, lds = s_all.shape
        need_w = [_need(ctx, 3 + i) for i in range(len(weights))]
        if torch.is_grad_enabled():
            # recorded again: DemodFn's formula layer by layer, on tensor ops
            gs_all = torch.zeros_like(s_all, dtype=torch.float32) if _need(ctx, 0) else None
            gws = []
            for i, (w, so, do) in enumerate(zip(weights, soffs, doffs)):
                cout, cin = w.shape[0], w.shape[1]
                sf, wf, g = s_all[:, so:so + cin].float(), w.float(), g_all[:, do:do + cout].float()
                wsq_t = wf.square().sum(dim=(2, 3))
                d = (sf.square() @ wsq_t.t()).rsqrt()
                dq = torch.where(d < 1e8, -0.5 * d * d * d * g, torch.zeros_like(d))
                if gs_all is not None:
                    gs_all = gs_all + F.pad(2.0 * sf * (dq @ wsq_t), (so, lds - so - cin))
                gws.append((2.0 * wf * (dq.t() @ sf.square())[:, :, None, None]).to(w.dtype) if need_w[i] else None)
            return (gs_all.to(s_all.dtype) if gs_all is not None else None), None, None, *gws
        lib = _lib.load()
        s32 = s_all.detach().float().contiguous()
        g32 = g_all.detach().float().contiguous()
        ldd = tables["ldd"]
        gs_all = None
        if _need(ctx, 0):
            gs_all = torch.zeros((n, lds), dtype=torch.float32, device=s32.device)  # (the columns of layers that do not demodulate stay 0)
            tab, blocks = tables["tabs"][1]
            check(lib.gan_demod_multi(1, ptr(tab), len(weights), blocks, ptr(s32), lds, ptr(g32), ptr(dinv_all), ldd, ptr(gs_all), n, stream_of(s32)),
                  "gan_demod_multi")
            gs_all = gs_all.to(s_all.dtype)
        gws = [None] * len(weights)
        if any(need_w):
            live = [grad_slot(w) for w in weights]
            if all(need_w) and all(g is not None and g.data_ptr() == q for g, q in zip(live, tables["slot_ptrs"])):
                tab, blocks = tables["tabs"][2]
                check(lib.gan_demod_multi(2, ptr(tab), len(weights), blocks, ptr(s32), lds, ptr(g32), ptr(dinv_all), ldd, None, n, stream_of(s32)),
                      "gan_demod_multi")
            else:  # (no in-place accumulation in this pass: one launch per layer, gradients handed to autograd)
                for i, (w, so, do, wsq, slot) in enumerate(zip(weights, soffs, doffs, tables["wsqs"], live)):
                    if not need_w[i]:
                        continue
                    cout, cin, kh, kw = w.shape
                    gw = slot if slot is not None else torch.empty_like(w, dtype=torch.float32)
                    gl, dl, sl = g32[:, do:do + cout].contiguous(), dinv_all[:, do:do + cout].contiguous(), s32[:, so:so + cin].contiguous()
                    check(lib.gan_demod_bwd(ptr(gl), ptr(dl), ptr(sl), ptr(w.detach()), ptr(wsq), None, ptr(gw), int(slot is not None), n, cin, cout,
                                            kh * kw, stream_of(s32)), "gan_demod_bwd")
                    gws[i] = None if slot is not None else gw.to(w.dtype)
        return gs_all, None, None, *gws


def demod_all(s_all: Tensor, soffs, weights):
    """The demodulation factor of every weight in `weights` (layer l's style = columns soffs[l] .. of s_all) as a tuple of
    contiguous (N, Cout_l) tensors from one launch, or None when the tables cannot be had (see _demod_tables): the caller
    then evaluates DemodFn layer by layer."""
    if not _FUSE_DEMOD or not weights or not s_all.is_cuda:
        return None
    tables = _demod_tables(weights, soffs, s_all.shape[0])
    if tables is None:
        # remembered: prepack_weights() builds the tables after the next optimizer step, when the tap sums exist and the
        # stream is not being captured (a trainer captures its second iteration: the first must leave them behind)
        if len(_DEMOD_WANTS) > 16:
            _DEMOD_WANTS.clear()
        _DEMOD_WANTS[(tuple(id(w) for w in weights), s_all.shape[0])] = ([weakref.ref(w) for w in weights], tuple(soffs), s_all.shape[0])
        return None
    dinv_all = DemodAllFn.apply(s_all, tuple(soffs), tables, *weights)
    return SplitColsFn.apply(dinv_all, tuple(w.shape[0] for w in weights))


class MbstdFn(Function):
    """(N/G,) fp32 statistic of MinibatchStdDev (progressive_gan.py:25-36): kernel gan_mbstd_fwd; the backward is
    MbstdBwdFn (kernel gan_mbstd_bwd), itself differentiable by kernel gan_mbstd_bwd2 -- R1 differentiates the
    discriminator twice (losses.py:100-103), and every node of that graph is a HIP kernel."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.save_for_backward(x)
        ctx.group = group
        return raw_mbstd_stat(x, group)

    @staticmethod
    def backward(ctx, gstat):
        (x,) = ctx.saved_tensors
        return MbstdBwdFn.apply(x, gstat, ctx.group), None


class MbstdBwdFn(Function):
    """gx[j + k*M] = gstat[j] * (x_k - mean_k x) / (G * std * C*H*W): the adjoint of the statistic in x, as a function
    of (x, gstat).  Its own adjoint (both arguments) is one launch of gan_mbstd_bwd2."""

    @staticmethod
    def forward(ctx, x, gstat, group):
        require_gpu(x, gstat)
        lib = _lib.load()
        xc = as_cl(x.detach())
        n, c, h, w = xc.shape
        g32 = _f32(gstat)
        gx = torch.empty_like(xc)
        check(lib.gan_mbstd_bwd(ptr(xc), ptr(g32), ptr(gx), n, group, h * w, c, dtype_code(xc), stream_of(xc)), "gan_mbstd_bwd")
        ctx.save_for_backward(xc, g32)
        ctx.group = group
        ctx.gstat_dtype = gstat.dtype
        return gx

    @staticmethod
    def backward(ctx, u):
        xc, g32 = ctx.saved_tensors
        group = ctx.group
        if torch.is_grad_enabled():
            # no caller: R1 / WGAN-GP (losses.py:89-103) stop at second order
            raise _lib.GanAmdError("MinibatchStdDev: derivatives beyond second order are not implemented")
        lib = _lib.load()
        n, c, h, w = xc.shape
        uc = as_cl(u.detach().to(xc.dtype))
        gx2 = torch.empty_like(xc) if _need(ctx, 0) else None
        ggstat = torch.empty((n // group,), dtype=torch.float32, device=xc.device) if _need(ctx, 1) else None
        if gx2 is None and ggstat is None:
            return None, None, None
        check(lib.gan_mbstd_bwd2(ptr(xc), ptr(g32), ptr(uc), ptr(gx2), ptr(ggstat), n, group, h * w, c, dtype_code(xc), stream_of(xc)),
              "gan_mbstd_bwd2")
        return gx2, (ggstat.to(ctx.gstat_dtype) if ggstat is not None else None), None


# ---- the image boundary: 1x1 convolution from / to <= 4 channels (include/gan_amd.h: gan_rgb_conv1x1_*) ----------
def _torch_dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return _lib.GAN_F32
    if dt == torch.bfloat16:
        return _lib.GAN_BF16
    raise _lib.GanAmdError(f"gan_amd kernels take float32 or bfloat16, got {dt}")


def rgb_conv_shape(x: Tensor, weight: Tensor) -> bool:
    """Shapes the image-boundary kernels serve: a 1x1 convolution of a <= 4-channel 4-D tensor into 8..64 channels."""
    cout, cin, kh, kw = weight.shape
    return (x.dim() == 4 and kh == 1 and kw == 1 and cin == x.shape[1] and 1 <= cin <= 4 and 8 <= cout <= 64 and cout % 8 == 0
            and x.dtype in (torch.float32, torch.bfloat16) and weight.dtype == torch.float32)


def _raw_rgb_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], slope: float, out_dtype: torch.dtype) -> Tensor:
    require_gpu(x, weight, bias)
    x = x.detach()
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    w32 = weight.detach().reshape(cout, cin).contiguous()
    b32 = _f32(bias)
    y = torch.empty((n, cout, h, w), dtype=out_dtype, device=x.device, memory_format=CL)
    sn, sc, sh, sw = x.stride()
    check(_run("rgb_conv1x1", 0, _nbytes(x, y),
               lambda: _lib.load().gan_rgb_conv1x1_fwd(ptr(x), dtype_code(x), sn, sc, sh, sw, ptr(w32), ptr(b32), ptr(y), _torch_dtype_code(out_dtype),
                                                       n, cin, h, w, cout, float(slope), stream_of(x)), (n, cin, h, w, cout)), "gan_rgb_conv1x1_fwd")
    return y


def _raw_rgb_dgrad(gy: Tensor, weight: Tensor) -> Tensor:
    require_gpu(gy, weight)
    gc = as_cl(gy.detach())
    n, cout, h, w = gc.shape
    cin = weight.shape[1]
    w32 = weight.detach().float().reshape(cout, cin).contiguous()
    gx = _empty_like_cl(gc, (n, cin, h, w))
    check(_run("rgb_conv1x1_dgrad", 0, _nbytes(gc, gx),
               lambda: _lib.load().gan_rgb_conv1x1_dgrad(ptr(gc), ptr(w32), ptr(gx), n * h * w, cin, cout, dtype_code(gc), stream_of(gc)), (n, cin, h, w, cout)),
          "gan_rgb_conv1x1_dgrad")
    return gx


def _raw_rgb_wgrad(x: Tensor, gy: Tensor, bias: bool = False):
    """dW (Cout, Cin, 1, 1) fp32; bias: -> (dW, dbias (Cout,)) from the same pass (Cin <= 3: a constant 1 in the free channel slot)."""
    require_gpu(x, gy)
    lib = _lib.load()
    x = x.detach()
    gc = as_cl(gy.detach())
    n, cin, h, w = x.shape
    cout = gc.shape[1]
    dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if bias else None
    ws_floats = lib.gan_rgb_conv1x1_wgrad_workspace(cin, cout)
    ws = torch.empty((ws_floats,), dtype=torch.float32, device=x.device)
    sn, sc, sh, sw = x.stride()
    check(_run("rgb_conv1x1_wgrad", 0, _nbytes(x, gc),
               lambda: lib.gan_rgb_conv1x1_wgrad(ptr(x), dtype_code(x), sn, sc, sh, sw, ptr(gc), dtype_code(gc), ptr(dw), ptr(db), ptr(ws), ws_floats, n, cin,
                                                 h, w, cout, stream_of(x)), (n, cin, h, w, cout)), "gan_rgb_conv1x1_wgrad")
    return (dw, db) if bias else dw


class RgbConvFn(Function):
    """y = lrelu_slope(conv1x1(x, W) + b) for an image-like x (<= 4 channels, any strides, fp32 or bf16): the discriminator's
    first layer (progressive_gan.py:61) without the padded copy.  RgbConvFn / RgbDgradFn / RgbWgradFn are closed under
    differentiation like the Conv2d family (the map is bilinear; slope = 1 at this layer in every caller)."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope, out_dtype):
        y = _raw_rgb_fwd(x, weight, bias, slope, out_dtype)
        ctx.save_for_backward(x, weight, bias, y if slope != 1.0 else None)
        ctx.slope = slope
        ctx.out_dtype = out_dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, y = ctx.saved_tensors
        gz = MaskMulFn.apply(gy, y, ctx.slope) if ctx.slope != 1.0 else gy
        gx = RgbDgradFn.apply(gz, weight).to(x.dtype) if _need(ctx, 0) else None
        if _need(ctx, 1) and bias is not None and _need(ctx, 2) and not torch.is_grad_enabled() and x.shape[1] < 4:
            gw, gb = _raw_rgb_wgrad(x, gz, bias=True)  # the bias gradient out of the weight gradient's pass over gz
            return gx, gw.to(weight.dtype), gb.reshape(bias.shape).to(bias.dtype), None, None
        gw = RgbWgradFn.apply(x, gz).to(weight.dtype) if _need(ctx, 1) else None
        gb = ChanSumFn.apply(gz, False).reshape(bias.shape).to(bias.dtype) if bias is not None and _need(ctx, 2) else None
        return gx, gw, gb, None, None


class RgbDgradFn(Function):
    """gx = gy . W over the output channels: the adjoint of RgbConvFn in x (channels-last, <= 4 channels, gy's dtype)."""

    @staticmethod
    def forward(ctx, gy, weight):
        ctx.save_for_backward(gy, weight)
        return _raw_rgb_dgrad(gy, weight)

    @staticmethod
    def backward(ctx, ggx):
        gy, weight = ctx.saved_tensors
        g_gy = RgbConvFn.apply(ggx, weight, None, 1.0, gy.dtype) if _need(ctx, 0) else None
        g_w = RgbWgradFn.apply(ggx, gy).to(weight.dtype) if _need(ctx, 1) else None
        return g_gy, g_w


class RgbWgradFn(Function):
    """dW = sum over samples and pixels of gy (x) x: the adjoint of RgbConvFn in W, fp32 (Cout, Cin, 1, 1)."""

    @staticmethod
    def forward(ctx, x, gy):
        ctx.save_for_backward(x, gy)
        return _raw_rgb_wgrad(x, gy)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        g_x = RgbDgradFn.apply(gy, ggw.float()).to(x.dtype) if _need(ctx, 0) else None
        g_gy = RgbConvFn.apply(x, ggw.float(), None, 1.0, gy.dtype) if _need(ctx, 1) else None
        return g_x, g_gy


def _raw_s2d(x: Tensor, pad: int, to_depth: bool, h: int, w: int, c: int) -> Tensor:
    """to_depth: x (N, C, H, W) -> (N, 4C, (H+2p)/2, (W+2p)/2); else x (N, 4C, Hb, Wb) -> (N, C, H, W) (h, w, c = the un-folded size)."""
    require_gpu(x)
    xc = as_cl(x.detach())
    n = xc.shape[0]
    if (h + 2 * pad) % 2 or (w + 2 * pad) % 2:
        raise _lib.GanAmdError(f"space_to_depth: padded size {h + 2 * pad}x{w + 2 * pad} must be even")
    shape = (n, 4 * c, (h + 2 * pad) // 2, (w + 2 * pad) // 2) if to_depth else (n, c, h, w)
    out = _empty_like_cl(xc, shape)
    check(_run("space_to_depth", 0, _nbytes(xc, out),
               lambda: _lib.load().gan_space_to_depth(ptr(xc), ptr(out), n, h, w, c, pad, int(to_depth), dtype_code(xc), stream_of(xc)), (n, c, h, w)),
          "gan_space_to_depth")
    return out


class SpaceToDepthFn(Function):
    """S2D_p: zero-pad by p, fold each 2x2 pixel block into the channels (channel (dy*2+dx)*C + c).  Linear; adjoint = DepthToSpaceFn."""

    @staticmethod
    def forward(ctx, x, pad):
        n, c, h, w = x.shape
        ctx.cfg = (pad, h, w, c)
        return _raw_s2d(x, pad, True, h, w, c)

    @staticmethod
    def backward(ctx, g):
        pad, h, w, c = ctx.cfg
        return DepthToSpaceFn.apply(g, pad, h, w), None


class DepthToSpaceFn(Function):
    """The adjoint of S2D_p: unfold the channels into 2x2 pixel blocks and crop p from every side -> (N, C, h, w)."""

    @staticmethod
    def forward(ctx, y, pad, h, w):
        c = y.shape[1] // 4
        ctx.pad = pad
        return _raw_s2d(y, pad, False, h, w, c)

    @staticmethod
    def backward(ctx, g):
        return SpaceToDepthFn.apply(g, ctx.pad), None, None, None


# (Co, C, 2m, 2m) stride-2 kernels as stride-1 kernels over space-to-depth input, once per version of the parameter
_FOLD_CACHE: dict = {}   # id(param) -> (weakref, (version, data_ptr), folded tensor)
_FOLD_OWNER: dict = {}   # data_ptr of a cached folded tensor -> weakref(param): lets pack_weight cache its GEMM images


def _fold_perm(w: Tensor) -> Tensor:
    co, c, k, _ = w.shape
    m = k // 2
    return w.reshape(co, c, m, 2, m, 2).permute(0, 3, 5, 1, 2, 4).reshape(co, 4 * c, m, m)


def _unfold_perm(g: Tensor) -> Tensor:
    co, c4, m, _ = g.shape
    c = c4 // 4
    return g.reshape(co, 2, 2, c, m, m).permute(0, 3, 4, 1, 5, 2).reshape(co, c, 2 * m, 2 * m)


class FoldWeightFn(Function):
    """(Co, C, 2m, 2m) -> (Co, 4C, m, m): input channel (dy*2+dx)*C + c, tap (by, bx) <- tap (2by+dy, 2bx+dx): the stride-2
    kernel of Conv2d / ConvTranspose2d (dcgan.py:21-29,53-61) as a stride-1 kernel over space-to-depth input.  A permutation
    of the parameter.  The folded copy of a PARAMETER is kept until its version moves (a WGAN-GP discriminator step used
    each weight ~8 times: 8 permute copies and 8 uncached packs), and pack_weight caches the GEMM images of that copy."""

    @staticmethod
    def forward(ctx, weight):
        if not (isinstance(weight, torch.nn.Parameter) and weight.is_leaf):
            return _fold_perm(weight.detach()).contiguous()
        ver = (weight._version, weight.data_ptr())
        hit = _FOLD_CACHE.get(id(weight))
        if hit is not None and hit[0]() is weight and hit[1] == ver:
            return hit[2].detach()  # a new alias per call: autograd writes its graph node on the tensor it is handed
        if len(_FOLD_CACHE) > 1024:
            _FOLD_CACHE.clear()
            _FOLD_OWNER.clear()
        if hit is not None:
            _FOLD_OWNER.pop(hit[2].data_ptr(), None)
        folded = _fold_perm(weight.detach()).contiguous()
        _FOLD_CACHE[id(weight)] = (weakref.ref(weight), ver, folded)
        _FOLD_OWNER[folded.data_ptr()] = weakref.ref(weight)
        return folded.detach()

    @staticmethod
    def backward(ctx, g):
        return _unfold_perm(g)


def _derived_owner(t: Tensor):
    """(kind, parameter, version key) when `t` aliases the current cached folded (FoldWeightFn) or scaled
    (ScaleConvParamsFn) copy of a parameter, else None."""
    p = _fold_owner(t) if _FOLD_OWNER else None
    if p is not None:
        return "fold", p, (p._version, p.data_ptr())
    ref = _SCALED_OWNER.get(t.data_ptr())
    p = ref() if ref is not None else None
    if p is None:
        return None
    hit = _SCALED_CACHE.get(id(p))
    if hit is None or hit[0]() is not p or hit[2][0].data_ptr() != t.data_ptr() or hit[2][0].shape != t.shape or hit[1] != hit[3]():
        return None
    return "scaled", p, hit[1]


def _fold_owner(t: Tensor):
    """The parameter whose current cached folded copy `t` aliases, or None."""
    ref = _FOLD_OWNER.get(t.data_ptr())
    p = ref() if ref is not None else None
    if p is None:
        return None
    hit = _FOLD_CACHE.get(id(p))
    if hit is None or hit[0]() is not p or hit[1] != (p._version, p.data_ptr()) or hit[2].data_ptr() != t.data_ptr() or hit[2].shape != t.shape:
        return None
    return p


class InstNormAffineFn(Function):
    """y = lrelu_slope( instance_norm(x) * gamma[n, c] + beta[n, c] ): AdaIN (modelling/common.py:13-16; F.instance_norm:
    biased variance over H x W, eps 1e-5) with the LeakyReLU that follows it in the StyleGAN generator block
    (stylegan.py:33-34).  Forward: statistics kernel + one affine pass; backward: two channel reductions + one pass
        gx = r * gamma * (gz - mean(gz) - xhat * mean(gz * xhat)),   gz = g * lrelu'(y),
    written as gz * a1 + x * a2 + a3 with per-(sample, channel) coefficients (kernel gan_chan_lin2).
    The kernels serve the first order (the generator never sits inside a create_graph region, train_celeba.py:128-129);
    under create_graph the backward is the same map out of differentiable library ops."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, eps):
        require_gpu(x, gamma, beta)
        lib = _lib.load()
        xc = as_cl(x.detach())
        n, c, h, w = xc.shape
        g32 = _f32(gamma).reshape(n, c)
        b32 = _f32(beta).reshape(n, c)
        mean = torch.empty((n, c), dtype=torch.float32, device=xc.device)
        rstd = torch.empty_like(mean)
        check(lib.gan_chan_stats(ptr(xc), ptr(mean), ptr(rstd), n, h * w, c, float(eps), dtype_code(xc), stream_of(xc)), "gan_chan_stats")
        a = rstd * g32
        b = b32 - mean * a
        y = torch.empty_like(xc)
        check(_run("chan_affine", 0, _nbytes(xc, y),
                   lambda: lib.gan_chan_affine(ptr(xc), ptr(a), ptr(b), ptr(y), n, h * w, c, float(slope), dtype_code(xc), stream_of(xc)), (n, c, h, w)),
              "gan_chan_affine")
        ctx.save_for_backward(xc, g32, mean, rstd, y if slope != 1.0 else None, x, gamma, beta)
        ctx.slope = slope
        ctx.eps = eps
        ctx.meta = (gamma.shape, gamma.dtype, beta.shape, beta.dtype, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, g32, mean, rstd, y, x_in, gamma_in, beta_in = ctx.saved_tensors
        if torch.is_grad_enabled():
            # a create_graph region (gradient penalties through a normalised network): the same map out of differentiable
            # library ops, so any order of derivative exists -- at library speed; the kernels serve the first order
            with torch.enable_grad():
                n, c = x_in.shape[:2]
                z = torch.nn.functional.instance_norm(x_in.float(), eps=ctx.eps) * gamma_in.float().reshape(n, c, 1, 1) + beta_in.float().reshape(n, c, 1, 1)
                z = torch.nn.functional.leaky_relu(z, ctx.slope) if ctx.slope != 1.0 else z
                wanted = [t for i, t in enumerate((x_in, gamma_in, beta_in)) if _need(ctx, i)]
                grads = iter(torch.autograd.grad(z, wanted, g.float(), create_graph=True, allow_unused=True))
            return tuple(next(grads) if _need(ctx, i) else None for i in range(3)) + (None, None)
        n, c, h, w = xc.shape
        hw = float(h * w)
        gc = as_cl(g.detach().to(xc.dtype))
        s1 = raw_chan_dot(gc, None, 0, y, ctx.slope, True)   # sum gz
        sx = raw_chan_dot(gc, xc, 1, y, ctx.slope, True)     # sum gz * x
        s2 = rstd * (sx - mean * s1)                         # sum gz * xhat
        gamma_shape, gamma_dtype, beta_shape, beta_dtype, x_dtype = ctx.meta
        ggamma = s2.reshape(gamma_shape).to(gamma_dtype) if _need(ctx, 1) else None
        gbeta = s1.reshape(beta_shape).to(beta_dtype) if _need(ctx, 2) else None
        gx = None
        if _need(ctx, 0):
            a1 = rstd * g32
            a2 = -(rstd * a1 * s2) / hw
            a3 = -(a1 * s1) / hw - a2 * mean
            gx = torch.empty_like(xc)
            check(_run("chan_lin2", 0, _nbytes(gc, xc, gx, y),
                       lambda: _lib.load().gan_chan_lin2(ptr(gc), ptr(y), ptr(xc), ptr(a1.contiguous()), ptr(a2.contiguous()), ptr(a3.contiguous()),
                                                         ptr(gx), n, h * w, c, float(ctx.slope), dtype_code(xc), stream_of(xc)), (n, c, h, w)),
                  "gan_chan_lin2")
            gx = gx.to(x_dtype)
        return gx, ggamma, gbeta, None, None


def instance_norm_affine(x: Tensor, gamma: Tensor, beta: Tensor, slope: float = 1.0, eps: float = 1e-5) -> Tensor:
    return InstNormAffineFn.apply(x, gamma, beta, slope, eps)


class LayerNorm2dFn(Function):
    """y = lrelu_slope( LayerNorm_over_channels(x) * w[c] + b[c] ) per pixel (modelling/common.py:19-24, eps 1e-5) with the
    LeakyReLU that follows it in the Progressive-GAN generator (progressive_gan.py:89-90,124-125): one launch forward,
    one backward (input gradient + both parameter gradients).  Under create_graph (a gradient penalty through
    DcGanDiscriminator(norm="layer")) the backward is the same map out of differentiable library ops."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope, eps):
        require_gpu(x, weight, bias)
        lib = _lib.load()
        xc = as_cl(x.detach())
        n, c, h, w = xc.shape
        w32, b32 = _f32(weight).reshape(c), _f32(bias).reshape(c)
        y = torch.empty_like(xc)
        stats = torch.empty((n * h * w, 2), dtype=torch.float32, device=xc.device)
        check(_run("layernorm2d", 0, _nbytes(xc, y),
                   lambda: lib.gan_layernorm2d_fwd(ptr(xc), ptr(w32), ptr(b32), ptr(y), ptr(stats), n * h * w, c, float(eps), float(slope),
                                                   dtype_code(xc), stream_of(xc)), (n, c, h, w)), "gan_layernorm2d_fwd")
        ctx.save_for_backward(xc, w32, stats, y if slope != 1.0 else None, x, weight, bias)
        ctx.slope = slope
        ctx.eps = eps
        ctx.meta = (weight.shape, weight.dtype, bias.shape, bias.dtype, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, w32, stats, y, x_in, w_in, b_in = ctx.saved_tensors
        if torch.is_grad_enabled():
            # a create_graph region (WGAN-GP / R1 through DcGanDiscriminator(norm="layer")): nn.LayerNorm over the channel
            # axis out of differentiable library ops, as the reference runs it (modelling/common.py:19-24)
            with torch.enable_grad():
                c = x_in.shape[1]
                z = torch.nn.functional.layer_norm(x_in.float().permute(0, 2, 3, 1), (c,), w_in.float().reshape(c), b_in.float().reshape(c), ctx.eps)
                z = z.permute(0, 3, 1, 2)
                z = torch.nn.functional.leaky_relu(z, ctx.slope) if ctx.slope != 1.0 else z
                wanted = [t for i, t in enumerate((x_in, w_in, b_in)) if _need(ctx, i)]
                grads = iter(torch.autograd.grad(z, wanted, g.float(), create_graph=True, allow_unused=True))
            return tuple(next(grads) if _need(ctx, i) else None for i in range(3)) + (None, None)
        n, c, h, w = xc.shape
        gc = as_cl(g.detach().to(xc.dtype))
        gx = torch.empty_like(xc) if _need(ctx, 0) else None
        gwb = torch.empty((2, c), dtype=torch.float32, device=xc.device)
        check(_run("layernorm2d_bwd", 0, _nbytes(gc, xc, gx, y),
                   lambda: _lib.load().gan_layernorm2d_bwd(ptr(gc), ptr(xc), ptr(y), ptr(w32), ptr(stats), ptr(gx), ptr(gwb), n * h * w, c,
                                                           float(ctx.slope), dtype_code(xc), stream_of(xc)), (n, c, h, w)), "gan_layernorm2d_bwd")
        wshape, wdtype, bshape, bdtype, x_dtype = ctx.meta
        return ((gx.to(x_dtype) if gx is not None else None), (gwb[0].reshape(wshape).to(wdtype) if _need(ctx, 1) else None),
                (gwb[1].reshape(bshape).to(bdtype) if _need(ctx, 2) else None), None, None)


def layer_norm2d(x: Tensor, weight: Tensor, bias: Tensor, slope: float = 1.0, eps: float = 1e-5) -> Tensor:
    return LayerNorm2dFn.apply(x, weight, bias, slope, eps)


class MbstdConcatFn(Function):
    """(x (N, C, H, W), stat (N/G,)) -> (N, C+1, H, W): the statistic of each sample's group column appended as a
    constant plane (progressive_gan.py:31-36), one launch.  Linear in both arguments; its adjoint is
    MbstdConcatAdjFn and vice versa, so the pair is closed under differentiation."""

    @staticmethod
    def forward(ctx, x, stat, group):
        require_gpu(x, stat)
        xc = as_cl(x.detach())
        n, c, h, w = xc.shape
        s32 = _f32(stat)
        out = _empty_like_cl(xc, (n, c + 1, h, w))
        check(_lib.load().gan_mbstd_concat(ptr(xc), ptr(s32), ptr(out), n, group, h * w, c, dtype_code(xc), stream_of(xc)), "gan_mbstd_concat")
        ctx.group = group
        ctx.meta = (x.dtype, stat.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        gx, gstat = MbstdConcatAdjFn.apply(g, ctx.group)
        return (gx.to(ctx.meta[0]) if _need(ctx, 0) else None), (gstat.to(ctx.meta[1]) if _need(ctx, 1) else None), None


class MbstdConcatAdjFn(Function):
    """g (N, C+1, H, W) -> (g[:, :C], sum of the plane channel over each group column's members and pixels)."""

    @staticmethod
    def forward(ctx, g, group):
        require_gpu(g)
        gc = as_cl(g.detach())
        n, co, h, w = gc.shape
        gx = _empty_like_cl(gc, (n, co - 1, h, w))
        gstat = torch.empty((n // group,), dtype=torch.float32, device=gc.device)
        check(_lib.load().gan_mbstd_concat_adj(ptr(gc), ptr(gx), ptr(gstat), n, group, h * w, co - 1, dtype_code(gc), stream_of(gc)),
              "gan_mbstd_concat_adj")
        ctx.group = group
        ctx.dtype = g.dtype
        return gx, gstat

    @staticmethod
    def backward(ctx, ggx, ggstat):
        return MbstdConcatFn.apply(ggx, ggstat, ctx.group).to(ctx.dtype), None


def minibatch_stddev(x: Tensor, group: int) -> Tensor:
    """MinibatchStdDev.forward (progressive_gan.py:25-36): (N, C, H, W) -> (N, C+1, H, W), two launches."""
    stat = minibatch_stddev_stat(x, group)
    return MbstdConcatFn.apply(x, stat, group)


def minibatch_stddev_stat(x: Tensor, group: int) -> Tensor:
    """(N/G,) fp32 statistic of MinibatchStdDev (progressive_gan.py:25-36)."""
    if not x.is_cuda:
        raise _lib.GanAmdError("gan_amd kernels need a CUDA/HIP tensor (no CPU path)")
    if not torch.is_grad_enabled() or not x.requires_grad:
        return raw_mbstd_stat(x, group)
    return MbstdFn.apply(x, group)
