"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement of the reference's StyleGAN2 hot path in plain PyTorch ops.
Every function takes explicit tensors / a flat ``state_dict``-style mapping whose keys
are the reference modules' parameter names, so the same parameters can be fed to the
reference (when generating goldens), to this oracle and to the HIP product modules.

All functions are dtype/device agnostic and differentiable to any order by ordinary
autograd (that is the point of an oracle: slow, obvious, trustworthy).

Citations are ``file:line`` in /root/reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

SLOPE = 0.2  # every LeakyReLU on the path: stylegan2.py:49,56,72 progressive_gan.py:44,46,73


# --------------------------------------------------------------------------------------
# modelling/nvidia_ops.py
# --------------------------------------------------------------------------------------
def blur_kernel(size: int) -> Tensor:
    """Binomial outer-product FIR, normalised to sum 1 (nvidia_ops.py:71-78)."""
    if not 2 <= size <= 7:
        raise AssertionError("kernel_size must be in [2, 7]")
    row = torch.tensor([math.comb(size - 1, i) for i in range(size)], dtype=torch.float32)
    row = row / row.sum()
    return torch.outer(row, row)


def upfirdn2d(imgs: Tensor, kernel: Tensor, up: int, down: int, px1: int, px2: int, py1: int, py2: int) -> Tensor:
    """Zero-insert x`up`, pad (negative = crop), correlate with the UNFLIPPED kernel,
    keep every `down`-th sample (nvidia_ops.py:18-34).  Written as an explicit tap
    loop instead of a depthwise conv so that it shares no code path with the product."""
    n, c, h, w = imgs.shape
    ky, kx = kernel.shape
    if up > 1:
        z = imgs.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = imgs
        imgs = z.reshape(n, c, h * up, w * up)
    imgs = F.pad(imgs, (px1, px2, py1, py2))
    hp, wp = imgs.shape[-2:]
    ho = (hp - ky) // down + 1
    wo = (wp - kx) // down + 1
    out = imgs.new_zeros(n, c, ho, wo)
    for i in range(ky):
        for j in range(kx):
            win = imgs[:, :, i : i + (ho - 1) * down + 1 : down, j : j + (wo - 1) * down + 1 : down]
            out = out + win * kernel[i, j].to(imgs.dtype)
    return out


def blur(imgs: Tensor, size: int = 3, up: int = 1, down: int = 1) -> Tensor:
    """`Blur(size, up, down)(imgs)` (nvidia_ops.py:58-69)."""
    p1 = (size - 1) // 2
    p2 = size - 1 - p1
    return upfirdn2d(imgs, blur_kernel(size).to(imgs.device, imgs.dtype if imgs.dtype == torch.float64 else torch.float32), up, down, p1, p2, p1, p2)


def nearest_up2(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2.0) / F.upsample(.., 2.0), mode nearest (stylegan2.py:45,99)."""
    return x.repeat_interleave(2, dim=-2).repeat_interleave(2, dim=-1)


def avg_pool2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(2) (progressive_gan.py:48,50)."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


def bias_leaky_relu(x: Tensor, bias: Optional[Tensor], slope: float = SLOPE) -> Tensor:
    """The fused op has no reference function; its oracle is the composition
    conv-bias -> LeakyReLU(0.2): no gain, no clamp (stylegan2.py:35,49 progressive_gan.py:44-47)."""
    if bias is not None:
        x = x + bias.view(1, -1, 1, 1).to(x.dtype)
    return F.leaky_relu(x, slope)


def up_conv_blur(sd: Mapping[str, Tensor], x: Tensor, kernel_size: int, blur_size: Optional[int] = 3) -> Tensor:
    """The nn.Sequential built by up_conv_blur (nvidia_ops.py:81-101), keys as in its state_dict:
    blur_size None: Upsample(2) -> Conv2d ("1.*"); kernel_size 1: Conv2d 1x1 ("0.*") -> Blur(up=2);
    else ConvTranspose2d(stride 2, output_padding k % 2) ("0.*") -> Blur."""
    pad = (kernel_size - 1) // 2
    if blur_size is None:
        return F.conv2d(nearest_up2(x), sd["1.weight"], sd["1.bias"], padding=pad)
    if kernel_size == 1:
        return blur(F.conv2d(x, sd["0.weight"], sd["0.bias"]), blur_size, up=2)
    y = F.conv_transpose2d(x, sd["0.weight"], sd["0.bias"], stride=2, padding=pad, output_padding=kernel_size % 2)
    return blur(y, blur_size)


def blur_conv_down(sd: Mapping[str, Tensor], x: Tensor, kernel_size: int, blur_size: Optional[int] = 3) -> Tensor:
    """The nn.Sequential built by blur_conv_down (nvidia_ops.py:104-123): blur_size None: Conv2d ("0.*") ->
    AvgPool2d(2); kernel_size 1: Blur(down=2) -> Conv2d 1x1 ("1.*"); else Blur -> Conv2d(stride 2) ("1.*")."""
    pad = (kernel_size - 1) // 2
    if blur_size is None:
        return avg_pool2(F.conv2d(x, sd["0.weight"], sd["0.bias"], padding=pad))
    if kernel_size == 1:
        return F.conv2d(blur(x, blur_size, down=2), sd["1.weight"], sd["1.bias"])
    return F.conv2d(blur(x, blur_size), sd["1.weight"], sd["1.bias"], stride=2, padding=pad)


# --------------------------------------------------------------------------------------
# modelling/stylegan2.py + modelling/stylegan.py:Noise
# --------------------------------------------------------------------------------------
# Test hook (tests/test_gpu_config4.py): emulate an implementation that STORES activations, their gradients and the
# convolution weights in a narrower dtype while computing in this oracle's own: with STORE_DTYPE set, those tensors are
# rounded to it (straight through: the gradient passes, itself rounded).  None (the default): the reference's arithmetic.
STORE_DTYPE = None


class _StoreRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _stored(x: Tensor) -> Tensor:
    return x if STORE_DTYPE is None else _StoreRound.apply(x, STORE_DTYPE)


def modulated_conv2d(
    x: Tensor, w: Tensor, weight: Tensor, bias: Tensor, style_weight: Tensor, style_bias: Tensor, demodulation: bool
) -> Tensor:
    """ModulatedConv2d.forward (stylegan2.py:29-36), literal form: per-sample weights,
    demodulated with w / max(||w||_2, 1e-8) over (Cin, kh, kw), grouped conv with groups=N."""
    n, cin, h, wd = x.shape
    cout, _, kh, kw = weight.shape
    style = F.linear(w, style_weight, style_bias) + 1  # (N, Cin)
    weight = _stored(weight)
    wmod = weight.unsqueeze(0) * style.view(n, 1, cin, 1, 1)  # (N, Cout, Cin, kh, kw)
    if demodulation:
        norm = wmod.square().sum(dim=(2, 3, 4), keepdim=True).sqrt()
        wmod = wmod / norm.clamp_min(1e-8)
    pad = (kh - 1) // 2
    y = F.conv2d(x.reshape(1, n * cin, h, wd).to(wmod.dtype), wmod.reshape(n * cout, cin, kh, kw), padding=pad, groups=n)
    y = y.reshape(n, cout, y.shape[-2], y.shape[-1])
    return y + bias.view(1, -1, 1, 1)


def add_noise(x: Tensor, scale: Tensor, noise: Optional[Tensor] = None) -> Tensor:
    """Noise.forward (stylegan.py:24-27): one randn(N,1,H,W) from the global RNG per call."""
    n, _, h, wd = x.shape
    if noise is None:
        noise = torch.randn(n, 1, h, wd, device=x.device, dtype=x.dtype)
    return x + noise.to(x.dtype) * scale.view(1, -1, 1, 1).to(x.dtype)


class _NoiseFeed:
    """Hands out recorded noise tensors in call order, or draws from the global RNG."""

    def __init__(self, recorded: Optional[Sequence[Tensor]]):
        self.recorded = list(recorded) if recorded is not None else None
        self.i = 0

    def next(self) -> Optional[Tensor]:
        if self.recorded is None:
            return None
        t = self.recorded[self.i]
        self.i += 1
        return t


def _sub(sd: Mapping[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix) :]: v for k, v in sd.items() if k.startswith(prefix)}


def _modconv_from(sd: Mapping[str, Tensor], prefix: str, x: Tensor, w: Tensor, demod: bool) -> Tensor:
    return modulated_conv2d(
        x,
        w,
        sd[prefix + "weight"],
        sd[prefix + "bias"],
        sd[prefix + "style_weight.weight"],
        sd[prefix + "style_weight.bias"],
        demod,
    )


def mapping_network(sd: Mapping[str, Tensor], z: Tensor) -> Tensor:
    """LayerNorm(z_dim) + 8 x (Linear, LeakyReLU 0.2) (stylegan2.py:70-72)."""
    h = F.layer_norm(z, z.shape[-1:], sd["mapping_network.0.weight"], sd["mapping_network.0.bias"])
    for i in range(8):
        k = 1 + 2 * i
        h = F.linear(h, sd[f"mapping_network.{k}.weight"], sd[f"mapping_network.{k}.bias"])
        h = F.leaky_relu(h, SLOPE)
    return h


def generator_block(
    sd: Mapping[str, Tensor], prefix: str, x: Tensor, w: Tensor, first: bool, feed: _NoiseFeed, blur_size: Optional[int] = None
):
    """StyleGan2GeneratorBlock.forward with the INTENDED to_rgb(x, w) (stylegan2.py:39-64).
    ModuleList indices: first block [modconv, noise, lrelu]; later blocks
    [upsample, modconv, noise, lrelu, modconv, noise, lrelu]."""
    if first:
        convs = [(0, 1)]
    elif blur_size is None:
        convs = [(1, 2), (4, 5)]
        x = nearest_up2(x)
    else:  # blur-enabled variant (no reference model: the `# TODO: blur` slot of stylegan2.py:47 filled with Blur)
        convs = [(1, 3), (5, 6)]
        x = nearest_up2(x)
    for j, (ci, ni) in enumerate(convs):
        x = _modconv_from(sd, f"{prefix}layers.{ci}.", x, w, True)
        if blur_size is not None and not first and j == 0:
            x = blur(x, blur_size)
        x = add_noise(x, sd[f"{prefix}layers.{ni}.scale"], feed.next())
        x = _stored(bias_leaky_relu(x, None))
    rgb = _stored(_modconv_from(sd, f"{prefix}to_rgb.", x, w, False))
    return x, rgb


def generator(sd: Mapping[str, Tensor], z: Tensor, noises: Optional[Sequence[Tensor]] = None, blur_size: Optional[int] = None) -> Tensor:
    """StyleGan2Generator.forward (stylegan2.py:92-101)."""
    feed = _NoiseFeed(noises)
    w = mapping_network(sd, z)
    n_blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    x = _stored(sd["learned_input"].expand(z.shape[0], -1, -1, -1))
    x, y = generator_block(sd, "blocks.0.", x, w, True, feed)
    for b in range(1, n_blocks):
        x, rgb = generator_block(sd, f"blocks.{b}.", x, w, False, feed, blur_size)
        y = _stored(nearest_up2(y) + rgb)
    return y


# --------------------------------------------------------------------------------------
# modelling/progressive_gan.py (the StyleGAN2 discriminator: modelling/__init__.py:37)
# --------------------------------------------------------------------------------------
def minibatch_stddev(imgs: Tensor, group_size: int = 4) -> Tensor:
    """MinibatchStdDev.forward (progressive_gan.py:25-36): groups {j + k*N/4}, population
    variance, eps 1e-6 inside the sqrt, mean over (C,H,W), one extra channel."""
    n, c, h, w = imgs.shape
    g = imgs.reshape(group_size, n // group_size, c, h, w)
    dev = g - g.mean(dim=0, keepdim=True)
    std = (dev.square().sum(dim=0) / group_size + 1e-6).sqrt()  # (M, C, H, W)
    stat = std.mean(dim=(1, 2, 3))  # (M,)
    plane = stat.repeat(group_size).view(n, 1, 1, 1).expand(n, 1, h, w)
    return torch.cat([imgs, plane.to(imgs.dtype)], dim=1)


def disc_block(sd: Mapping[str, Tensor], prefix: str, x: Tensor, blur_size: Optional[int] = None) -> Tensor:
    """ProgressiveGanDiscriminatorBlock.forward (progressive_gan.py:40-54), out-of-place LeakyReLU.  blur_size: both
    AvgPool2d(2) replaced by Blur(blur_size, down=2) (the pairing of blur_conv_down, nvidia_ops.py:104-123)."""
    down = avg_pool2 if blur_size is None else (lambda t: blur(t, blur_size, down=2))
    h = bias_leaky_relu(x, None)
    h = F.conv2d(h, sd[prefix + "layers.1.weight"], sd[prefix + "layers.1.bias"], padding=1)
    h = bias_leaky_relu(h, None)
    h = F.conv2d(h, sd[prefix + "layers.3.weight"], sd[prefix + "layers.3.bias"], padding=1)
    h = down(h)
    s = F.conv2d(down(x), sd[prefix + "shortcut.1.weight"], sd[prefix + "shortcut.1.bias"])
    return s + h * sd[prefix + "scale"]


def discriminator(sd: Mapping[str, Tensor], imgs: Tensor, blur_size: Optional[int] = None) -> Tensor:
    """ProgressiveGanDiscriminator (progressive_gan.py:57-82) -> logits (N,)."""
    last = max(int(k.split(".")[0]) for k in sd if k.split(".")[0].isdigit())
    x = F.conv2d(imgs, sd["0.weight"], sd["0.bias"])
    for b in range(1, last):
        x = disc_block(sd, f"{b}.", x, blur_size)
    x = minibatch_stddev(x)
    x = F.conv2d(x, sd[f"{last}.1.weight"], sd[f"{last}.1.bias"], padding=1)
    x = bias_leaky_relu(x, None)
    x = F.conv2d(x, sd[f"{last}.3.weight"], sd[f"{last}.3.bias"])
    return x.flatten()


# --------------------------------------------------------------------------------------
# modelling/stylegan.py (StyleGAN v1 generator) + modelling/progressive_gan.py:85-133 (Progressive GAN generator)
# + modelling/common.py:7-24 (AdaIN, LayerNorm2d)
# --------------------------------------------------------------------------------------
def adain(x: Tensor, w: Tensor, sw_w: Tensor, sw_b: Tensor, sb_w: Tensor, sb_b: Tensor) -> Tensor:
    """AdaIN.forward (common.py:13-16): instance norm (biased variance over H x W, eps 1e-5) * (style_w(w) + 1) + style_b(w)."""
    weight = (F.linear(w, sw_w, sw_b) + 1)[:, :, None, None]
    bias = F.linear(w, sb_w, sb_b)[:, :, None, None]
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = (x - mean).square().mean(dim=(2, 3), keepdim=True)
    return (x - mean) / (var + 1e-5).sqrt() * weight + bias


def layer_norm2d(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """LayerNorm2d.forward (common.py:20-24): LayerNorm over the channels of each pixel, eps 1e-5."""
    mean = x.mean(dim=1, keepdim=True)
    var = (x - mean).square().mean(dim=1, keepdim=True)
    return (x - mean) / (var + 1e-5).sqrt() * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def stylegan_generator(sd: Mapping[str, Tensor], z: Tensor, noises: Optional[Sequence[Tensor]] = None) -> Tensor:
    """StyleGanGenerator.forward (stylegan.py:76-85); block = [AdaIN, LeakyReLU, Upsample, Conv, Noise, LeakyReLU, Conv, Noise]
    (stylegan.py:30-47); w = mapping_network(z) * 0.01."""
    feed = _NoiseFeed(noises)
    w = mapping_network(sd, z) * 0.01
    n_blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    x = sd["learned_input"].expand(z.shape[0], -1, -1, -1)
    out = None
    for b in range(n_blocks):
        p = f"blocks.{b}."
        x = adain(x, w, sd[p + "0.style_w.weight"], sd[p + "0.style_w.bias"], sd[p + "0.style_b.weight"], sd[p + "0.style_b.bias"])
        x = nearest_up2(F.leaky_relu(x, SLOPE))
        x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
        x = F.leaky_relu(add_noise(x, sd[p + "4.scale"], feed.next()), SLOPE)
        x = F.conv2d(x, sd[p + "6.weight"], sd[p + "6.bias"], padding=1)
        x = add_noise(x, sd[p + "7.scale"], feed.next())
        rgb = F.conv2d(x, sd[f"out_convs.{b}.weight"], sd[f"out_convs.{b}.bias"])
        out = rgb if out is None else nearest_up2(out) + rgb
    return out


def progan_generator(sd: Mapping[str, Tensor], z: Tensor) -> Tensor:
    """ProgressiveGanGenerator (progressive_gan.py:103-133) with its blocks (progressive_gan.py:85-100)."""
    last = max(int(k.split(".")[0]) for k in sd)
    h = F.layer_norm(z, z.shape[-1:], sd["0.0.weight"], sd["0.0.bias"])
    x = F.conv_transpose2d(h[:, :, None, None], sd["0.2.weight"], sd["0.2.bias"])
    for b in range(1, last):
        p = f"{b}."
        h = F.leaky_relu(layer_norm2d(x, sd[p + "layers.0.weight"], sd[p + "layers.0.bias"]), SLOPE)
        h = F.conv2d(nearest_up2(h), sd[p + "layers.3.weight"], sd[p + "layers.3.bias"], padding=1)
        h = F.conv2d(F.leaky_relu(h, SLOPE), sd[p + "layers.5.weight"], sd[p + "layers.5.bias"], padding=1)
        s = nearest_up2(F.conv2d(x, sd[p + "shortcut.0.weight"], sd[p + "shortcut.0.bias"]))
        x = s + h * sd[p + "scale"]
    x = F.leaky_relu(layer_norm2d(x, sd[f"{last}.0.weight"], sd[f"{last}.0.bias"]), SLOPE)
    return F.conv2d(x, sd[f"{last}.2.weight"], sd[f"{last}.2.bias"])


# --------------------------------------------------------------------------------------
# modelling/dcgan.py + modelling/common.py (BASELINE config 5)
# --------------------------------------------------------------------------------------
def dcgan_discriminator(sd: Mapping[str, Tensor], imgs: Tensor) -> Tensor:
    """DcGanDiscriminator(norm="none") (dcgan.py:16-33, common.py:46-70): 4x4 stride-2 convs +
    LeakyReLU(0.2), then a valid 4x4 conv to one logit."""
    last = max(int(k.split(".")[0]) for k in sd)
    x = imgs
    for i in range(last):
        x = F.conv2d(x, sd[f"{i}.0.weight"], sd[f"{i}.0.bias"], stride=2, padding=1)
        x = bias_leaky_relu(x, None)
    x = F.conv2d(x, sd[f"{last}.weight"], sd[f"{last}.bias"])
    return x.flatten()


def dcgan_generator(sd: Mapping[str, Tensor], z: Tensor) -> Tensor:
    """DcGanGenerator(norm="batch") (dcgan.py:36-62): ConvT(z,4) + BN(batch stats, common.py:30) + ReLU,
    ConvT 4x4 s2 + BN + ReLU ..., ConvT 4x4 s2 (bias) + Tanh."""
    last = max(int(k.split(".")[0]) for k in sd)
    x = z.view(z.shape[0], -1, 1, 1)
    x = F.conv_transpose2d(x, sd["0.1.weight"], sd["0.1.bias"])
    x = F.batch_norm(x, None, None, sd["0.2.weight"], sd["0.2.bias"], training=True)
    x = x.relu()
    for i in range(1, last):
        x = F.conv_transpose2d(x, sd[f"{i}.0.weight"], stride=2, padding=1)
        x = F.batch_norm(x, None, None, sd[f"{i}.1.weight"], sd[f"{i}.1.bias"], training=True)
        x = x.relu()
    x = F.conv_transpose2d(x, sd[f"{last}.0.weight"], sd[f"{last}.0.bias"], stride=2, padding=1)
    return x.tanh()


# --------------------------------------------------------------------------------------
# losses.py
# --------------------------------------------------------------------------------------
def diff_augment(x: Tensor, brightness: Tensor, saturation: Tensor, contrast: Tensor, translate_x: int, translate_y: int,
                 offset_y: int, offset_x: int) -> Tensor:
    """DiffAugment (diff_augment.py:54-60) with its random draws made explicit: `brightness` / `saturation` / `contrast` are
    the torch.rand(N, 1, 1, 1) values of :14, :19, :24; the integers those of :42 (x, then y) and :48-49 (y, then x)."""
    n = x.shape[0]
    b, s, k = (t.reshape(n, 1, 1, 1).to(x.dtype) for t in (brightness, saturation, contrast))
    x = x + b - 0.5                                            # rand_brightness :13-14
    mean = x.mean(dim=-3, keepdim=True)
    x = mean + (x - mean) * s * 2                              # rand_saturation :17-19
    mean = x.mean(dim=[-1, -2, -3], keepdim=True)
    x = mean + (x - mean) * (k + 0.5)                          # rand_contrast :22-24
    x = F.pad(x, (translate_x, -translate_x, translate_y, -translate_y))  # translate :35-36
    h = x.shape[-2]
    mask = torch.ones_like(x)                                  # RandomCutout :45-51 (H // 2 for both sides, as written there)
    mask[..., offset_y : offset_y + h // 2, offset_x : offset_x + h // 2] = 0
    return x * mask


def d_loss(method: str, d_reals: Tensor, d_fakes: Tensor) -> Tensor:
    """losses.py:27-83, discriminator side."""
    if method == "gan":
        return -F.logsigmoid(d_reals).mean() - F.logsigmoid(-d_fakes).mean()
    if method == "lsgan":
        return (d_reals - 1).square().mean() + d_fakes.square().mean()
    if method == "wgan":
        return d_fakes.mean() - d_reals.mean()
    if method == "hinge":
        return (1 - d_reals).clamp_min(0).mean() + (1 + d_fakes).clamp_min(0).mean()
    if method == "relativistic-gan":
        return -F.logsigmoid(d_reals - d_fakes).mean()
    raise KeyError(method)


def g_loss(method: str, d_fakes: Tensor, d_reals_fn: Callable[[], Tensor]) -> Tensor:
    """losses.py:27-83, generator side (hinge shares the WGAN form, losses.py:69)."""
    if method == "gan":
        return -F.logsigmoid(d_fakes).mean()
    if method == "lsgan":
        return (d_fakes - 1).square().mean()
    if method in ("wgan", "hinge"):
        return -d_fakes.mean()
    if method == "relativistic-gan":
        with torch.no_grad():
            d_reals = d_reals_fn()
        return -F.logsigmoid(d_fakes - d_reals).mean()
    raise KeyError(method)


def r1_penalty(disc: Callable[[Tensor], Tensor], reals: Tensor, fakes: Tensor, d_reals: Tensor) -> Tensor:
    """r1_regularizer (losses.py:100-103): 5 * sum(grad^2) / N, create_graph."""
    (g,) = torch.autograd.grad(d_reals.sum(), reals, create_graph=True)
    return g.square().sum() / g.shape[0] * 5


def wgan_gp_penalty(
    disc: Callable[[Tensor], Tensor], reals: Tensor, fakes: Tensor, d_reals: Tensor, alpha: Optional[Tensor] = None
) -> Tensor:
    """wgan_gp_regularizer (losses.py:89-96); alpha ~ U(0,1) per sample from the global RNG unless given."""
    if alpha is None:
        alpha = torch.rand(fakes.shape[0], 1, 1, 1, device=fakes.device, dtype=fakes.dtype)
    inter = reals.detach().to(fakes.dtype).lerp(fakes.detach(), alpha).requires_grad_()
    d_inter = disc(inter)
    (g,) = torch.autograd.grad(d_inter.sum(), inter, create_graph=True)
    norm = g.flatten(1).square().sum(dim=1).sqrt()
    return (norm - 1).square().mean() * 10


def regularizer(name: str):
    return {"none": None, "r1": r1_penalty, "wgan-gp": wgan_gp_penalty}[name]


# --------------------------------------------------------------------------------------
# utils.py:EMA
# --------------------------------------------------------------------------------------
def ema_step(ema_params: Sequence[Tensor], params: Sequence[Tensor], counter: int, beta: float = 0.999, warmup: int = 100) -> int:
    """EMA.step (utils.py:84-96): counter += 1; nothing before `warmup`; copy at `warmup`; lerp after.
    Parameters only (buffers are not averaged)."""
    counter += 1
    with torch.no_grad():
        if counter < warmup:
            return counter
        if counter == warmup:
            for e, p in zip(ema_params, params):
                e.copy_(p)
            return counter
        for e, p in zip(ema_params, params):
            e.lerp_(p, 1 - beta)
    return counter


# --------------------------------------------------------------------------------------
# train_celeba.py:111-149 -- one generator iteration
# --------------------------------------------------------------------------------------
def init_like_reference(sd: Dict[str, Tensor], kind: str, generator: Optional[torch.Generator] = None) -> None:
    """In-place initialisation with the reference's rules.  kind="progressive": kaiming-normal
    (fan_in, gain sqrt(2)) on every conv/linear weight, zero bias, everything else untouched
    (progressive_gan.py:136-140).  kind="dcgan": N(0,0.02) conv weights, N(1,0.02) BN weights,
    zero biases (dcgan.py:65-72)."""
    for k, v in sd.items():
        with torch.no_grad():
            if kind == "progressive":
                if k.endswith("weight") and v.ndim >= 2:
                    fan_in = v[0].numel()
                    v.normal_(0, math.sqrt(2.0 / fan_in), generator=generator)
                elif k.endswith("bias") and not k.startswith("mapping_network.0."):
                    v.zero_()
            else:
                if k.endswith("weight") and v.ndim >= 2:
                    v.normal_(0, 0.02, generator=generator)
                elif k.endswith("weight"):
                    v.normal_(1, 0.02, generator=generator)
                elif k.endswith("bias"):
                    v.zero_()


def stylegan2_generator_shapes(img_size: int, img_channels: int = 3, z_dim: int = 128, base_dim: int = 16) -> Dict[str, tuple]:
    """Parameter names/shapes of StyleGan2Generator (stylegan2.py:67-90)."""
    shapes: Dict[str, tuple] = {"learned_input": (1, 512, 4, 4)}
    shapes["mapping_network.0.weight"] = (z_dim,)
    shapes["mapping_network.0.bias"] = (z_dim,)
    for i in range(8):
        shapes[f"mapping_network.{1 + 2 * i}.weight"] = (z_dim, z_dim)
        shapes[f"mapping_network.{1 + 2 * i}.bias"] = (z_dim,)

    def modconv(prefix, cin, cout, k):
        shapes[prefix + "weight"] = (cout, cin, k, k)
        shapes[prefix + "bias"] = (cout,)
        shapes[prefix + "style_weight.weight"] = (cin, z_dim)
        shapes[prefix + "style_weight.bias"] = (cin,)

    out_ch = min(base_dim * img_size // 4, 512)
    modconv("blocks.0.layers.0.", out_ch, out_ch, 3)
    shapes["blocks.0.layers.1.scale"] = (out_ch, 1, 1)
    modconv("blocks.0.to_rgb.", out_ch, img_channels, 1)
    in_ch = out_ch
    depth = int(math.log2(img_size // 4))
    for i in range(depth):
        out_ch = min(base_dim * img_size // 4 // 2 ** (i + 1), 512)
        p = f"blocks.{i + 1}."
        modconv(p + "layers.1.", in_ch, out_ch, 3)
        shapes[p + "layers.2.scale"] = (out_ch, 1, 1)
        modconv(p + "layers.4.", out_ch, out_ch, 3)
        shapes[p + "layers.5.scale"] = (out_ch, 1, 1)
        modconv(p + "to_rgb.", out_ch, img_channels, 1)
        in_ch = out_ch
    return shapes


def stylegan2_discriminator_shapes(img_size: int, img_channels: int = 3, base_dim: int = 16) -> Dict[str, tuple]:
    """Parameter names/shapes of ProgressiveGanDiscriminator (progressive_gan.py:57-77)."""
    shapes: Dict[str, tuple] = {"0.weight": (base_dim, img_channels, 1, 1), "0.bias": (base_dim,)}
    depth = int(math.log2(img_size // 4))
    in_ch = base_dim
    for b in range(1, depth + 1):
        out_ch = min(in_ch * 2, 512)
        shapes[f"{b}.scale"] = (out_ch, 1, 1)
        shapes[f"{b}.layers.1.weight"] = (in_ch, in_ch, 3, 3)
        shapes[f"{b}.layers.1.bias"] = (in_ch,)
        shapes[f"{b}.layers.3.weight"] = (out_ch, in_ch, 3, 3)
        shapes[f"{b}.layers.3.bias"] = (out_ch,)
        shapes[f"{b}.shortcut.1.weight"] = (out_ch, in_ch, 1, 1)
        shapes[f"{b}.shortcut.1.bias"] = (out_ch,)
        in_ch = out_ch
    last = depth + 1
    shapes[f"{last}.1.weight"] = (in_ch, in_ch + 1, 3, 3)
    shapes[f"{last}.1.bias"] = (in_ch,)
    shapes[f"{last}.3.weight"] = (1, in_ch, 4, 4)
    shapes[f"{last}.3.bias"] = (1,)
    return shapes


def new_stylegan2_state(img_size: int, gen_base_dim: int, disc_base_dim: int, device="cpu", dtype=torch.float32):
    """Fresh (generator, discriminator) parameter dicts initialised like the reference modules:
    learned_input = 1, Noise.scale = 0, D block scale = 1e-4, LayerNorm default, the rest init_weights."""
    g = {k: torch.zeros(s, device=device, dtype=dtype) for k, s in stylegan2_generator_shapes(img_size, base_dim=gen_base_dim).items()}
    d = {k: torch.zeros(s, device=device, dtype=dtype) for k, s in stylegan2_discriminator_shapes(img_size, base_dim=disc_base_dim).items()}
    init_like_reference(g, "progressive")
    init_like_reference(d, "progressive")
    with torch.no_grad():
        g["learned_input"].fill_(1.0)
        g["mapping_network.0.weight"].fill_(1.0)
        g["mapping_network.0.bias"].zero_()
        for k in g:
            if k.endswith(".scale"):
                g[k].zero_()
        for k in d:
            if k.endswith(".scale"):
                d[k].fill_(1e-4)
    for t in list(g.values()) + list(d.values()):
        t.requires_grad_(True)
    return g, d


def train_iteration(
    g_sd: Dict[str, Tensor],
    d_sd: Dict[str, Tensor],
    optim_g: torch.optim.Optimizer,
    optim_d: torch.optim.Optimizer,
    reals_fn: Callable[[], Tensor],
    *,
    method: str = "hinge",
    reg: str = "r1",
    n_disc: int = 1,
    z_dim: int = 128,
    gen_fn: Callable = generator,
    disc_fn: Callable = discriminator,
) -> Dict[str, float]:
    """One `while` body of train_celeba.py:111-149 with grad_accum=1, no autocast, no DiffAugment:
    n_disc discriminator updates (loss + regulariser every step), then one generator update with the
    discriminator frozen.  Random draws follow the reference's call order: z, [noise...], [alpha]."""
    reg_fn = regularizer(reg)
    disc = lambda imgs: disc_fn(d_sd, imgs)  # noqa: E731
    d_losses = []
    for i in range(n_disc):
        if method == "wgan" and reg == "none":
            with torch.no_grad():
                for p in d_sd.values():
                    p.clamp_(-0.01, 0.01)
        reals = reals_fn()
        cached = reals.clone()
        reals.requires_grad_()
        zs = torch.randn(reals.shape[0], z_dim, device=reals.device)
        with torch.no_grad():
            fakes = gen_fn(g_sd, zs)
        d_reals = disc(reals)
        d_fakes = disc(fakes)
        loss_d = d_loss(method, d_reals, d_fakes)
        if reg_fn is not None:
            loss_d = loss_d + reg_fn(disc, reals, fakes, d_reals)
        loss_d.backward()
        optim_d.step()
        optim_d.zero_grad()
        d_losses.append(float(loss_d.detach()))

    for p in d_sd.values():
        p.requires_grad_(False)
    zs = torch.randn(cached.shape[0], z_dim, device=cached.device)
    loss_g = g_loss(method, disc(gen_fn(g_sd, zs)), lambda: disc(cached))
    loss_g.backward()
    optim_g.step()
    optim_g.zero_grad()
    for p in d_sd.values():
        p.requires_grad_(True)
    return {
        "loss_d": float(loss_d.detach()),
        "loss_g": float(loss_g.detach()),
        "d_real": float(d_reals.detach().mean()),
        "d_fake": float(d_fakes.detach().mean()),
        "loss_d_all": d_losses,
    }
