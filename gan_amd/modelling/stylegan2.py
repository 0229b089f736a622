"""StyleGAN2 generator on the MFMA modulated-convolution kernel.

Reference: modelling/stylegan2.py (same class names, constructor signatures, forward signatures and
state_dict keys).  What differs is how the arithmetic is organised for MI355X:

* ModulatedConv2d never materialises the per-sample weights W[n] = W * s[n] / ||W * s[n]|| that the
  reference builds (stylegan2.py:30-33).  The identical result is computed as
      y[n] = dinv[n, co] * conv(x[n] * s[n, ci], W) + b,
      dinv[n, co] = 1 / max( sqrt( sum_ci s[n,ci]^2 * sum_k W[co,ci,k]^2 ), 1e-8 )
  i.e. one shared-weight implicit GEMM (bf16 MFMA, fp32 accumulate) with the modulation applied to
  the activation tile while it is staged into LDS, and demodulation + bias + noise + leaky-ReLU
  applied to the accumulators before the single store (kernel gan_conv2d_fwd).
* nn.Upsample(2) in front of a block's first conv (stylegan2.py:45) is folded into that conv's
  input gather (`up2`): the 4x larger tensor is never written.
* `to_rgb(x, w_embs)`: the shipped block calls `self.to_rgb(x)` without the style (stylegan2.py:64,
  a TypeError); the intended call is implemented.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import functional as GF
from ..functional import Function
from .common import LeakyReLU, Upsample, to_act
from .progressive_gan import init_weights
from .stylegan import Noise


def batched_conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, stride: int = 1, padding: int = 0) -> Tensor:
    """The reference's `batched_conv2d = torch.vmap(F.conv2d)` (stylegan2.py:18): one convolution per sample with that
    sample's own weights -- x (N, Cin, H, W), weight (N, Cout, Cin, kh, kw), bias (N, Cout) or None.  Kept for API
    parity: ModulatedConv2d below never forms per-sample weights (module docstring), so this entry is off the hot
    path and simply runs the shared-weight implicit-GEMM kernel once per sample; differentiable to any order in x,
    weight and bias (Conv2dFn)."""
    if stride not in (1, (1, 1)):
        raise GF._lib.GanAmdError("batched_conv2d supports stride 1")
    if x.dim() != 4 or weight.dim() != 5 or weight.shape[0] != x.shape[0]:
        raise GF._lib.GanAmdError(f"batched_conv2d: x {tuple(x.shape)} and weight {tuple(weight.shape)} do not pair up per sample")
    py, px = (padding, padding) if isinstance(padding, int) else padding
    x = to_act(x)
    ys = [GF.Conv2dFn.apply(x[n:n + 1], weight[n], py, px) for n in range(x.shape[0])]
    y = torch.cat(ys, dim=0)
    if bias is not None:
        y = y + bias.to(y.dtype).view(x.shape[0], -1, 1, 1)
    return y


class ModConvFn(Function):
    """y = lrelu_slope( dinv * conv(U(x) * s, W) + b + noise * ns ),  U = nearest x2 if up2.
    s, dinv, b, noise / ns may each be None: with s = dinv = None this is the plain fused
    [Upsample ->] Conv2d -> Noise -> LeakyReLU of the StyleGAN-v1 / Progressive-GAN generator blocks
    (stylegan.py:35-38, progressive_gan.py:91-92).

    Forward: one fused kernel.  Backward without create_graph: hand-written kernels (leaky-ReLU
    mask + demod scale in one pass, channel reductions for b / ns / dinv, dgrad, wgrad with the
    modulation fused into its operand staging).  Backward under create_graph: the same forward is
    re-expressed through the differentiable primitives of gan_amd.functional and differentiated, so
    any order of gradient stays on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, s, dinv, bias, noise, ns, slope, up2):
        pad = (weight.shape[2] - 1) // 2
        y = GF.raw_conv2d(
            x, weight, pad, pad, up2=up2, in_scale=s, out_scale=dinv, bias=bias, noise=noise, noise_scale=ns, slope=slope
        )
        ctx.save_for_backward(x, weight, s, dinv, bias, noise, ns, y)
        ctx.slope, ctx.up2, ctx.pad = slope, up2, pad
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, s, dinv, bias, noise, ns, y = ctx.saved_tensors
        slope, up2, pad = ctx.slope, ctx.up2, ctx.pad
        if torch.is_grad_enabled():
            return ModConvFn._backward_differentiable(ctx, gy)
        need = ctx.needs_input_grad
        gy = GF.as_cl(gy)
        n, cout = gy.shape[0], gy.shape[1]
        kh = weight.shape[2]
        # ---- one pass over (gy, y, noise): gz = gy * m(y) * dinv, the gradient entering the shared-weight
        # convolution, and the three per-sample channel sums behind the bias / noise-scale / demodulation
        # gradients ------------------------------------------------------------------------------------
        gb = gns = gdinv = None
        gz, sums = GF.raw_modconv_bwd_pre(gy, y if slope != 1.0 else None, noise if ns is not None else None, dinv, slope)
        if slope == 1.0 and dinv is not None and need[3]:  # no activation: the kernel had no y to form sum gy*y with
            sums[:, 2] = GF.raw_chan_dot(gy, y, 1, None, 1.0, True)
        want_gb = bias is not None and need[4]
        want_gns = ns is not None and need[6]
        want_gdinv = dinv is not None and need[3]
        if want_gb or want_gns or want_gdinv:
            # z*dinv = lrelu^-1(y) - b - noise*ns   and   gy*m(y)*lrelu^-1(y) = gy*y
            gb, gns, gdinv = GF.raw_modconv_bwd_post(sums, bias, ns if ns is not None else None, dinv, want_gb, want_gns, want_gdinv)
            if gb is not None:
                gb = gb.reshape(bias.shape).to(bias.dtype)
            if gns is not None:
                gns = gns.reshape(ns.shape).to(ns.dtype)
            if gdinv is not None:
                gdinv = gdinv.reshape(dinv.shape).to(dinv.dtype)
        gx = gw = gs = None
        if need[0] or need[2]:
            # d/d(U(x)*s); with U = nearest x2 its adjoint (the sum of each 2x2 block) rides on the accumulators
            tpad = kh - 1 - pad
            fused = None
            if need[0] and need[2]:
                # thin layers: dx = gxs * s and ds = sum_hw gxs * x on the convolution's accumulators (no gxs tensor)
                fused = GF.raw_conv2d_dgrad_scale_dot(gz, weight, tpad, tpad, s, x, pool=up2, gain=4.0 if up2 else 1.0)
            if fused is not None:
                gx, gs = fused
                gs = gs.reshape(s.shape).to(s.dtype)
            else:
                gxs = GF.raw_conv2d(gz, weight, tpad, tpad, transposed=True, pool=up2, gain=4.0 if up2 else 1.0)
                if need[0] and need[2]:
                    gx, gs = GF.raw_scale_dot(gxs, x, s)  # one pass: dx = gxs * s and ds = sum_hw gxs * x
                    gs = gs.reshape(s.shape).to(s.dtype)
                elif need[2]:
                    gs = GF.raw_chan_dot(gxs, x, 1, None, 1.0, True).reshape(s.shape).to(s.dtype)
                else:
                    gx = GF.raw_mask_scale(gxs, None, s, 1.0) if s is not None else gxs
        if need[1]:
            slot = GF.grad_slot(weight)  # a buffer the fold kernel adds to (functional.py: direct_param_grads), or None
            gw = GF.raw_conv2d_wgrad(x, gz, kh, kh, pad, pad, up2=up2, in_scale=s, into=slot)
            gw = None if slot is not None else gw.to(weight.dtype)
        return gx, gw, gs, gdinv, gb, None, gns, None, None

    @staticmethod
    def _backward_differentiable(ctx, gy):
        """The same gradients as `backward`, written as a composition of gan_amd.functional's
        differentiable primitives applied to the saved tensors and gy, so autograd records a graph of
        HIP-kernel Functions it can differentiate again (each primitive is local: no gradient flows
        through the history of s / dinv here, the outer graph does that)."""
        x, weight, s, dinv, bias, noise, ns, y = ctx.saved_tensors
        slope, up2, pad = ctx.slope, ctx.up2, ctx.pad
        need = ctx.needs_input_grad
        kh, kw = weight.shape[2], weight.shape[3]
        gp = GF.MaskMulFn.apply(gy, y, slope) if slope != 1.0 else gy  # gradient at the pre-activation
        xu = GF.nearest_up2(x) if up2 else x
        xs = GF.ChanScaleFn.apply(xu, s) if s is not None else xu
        gx = gw = gs = gdinv = gb = gns = None
        if bias is not None and need[4]:
            gb = GF.ChanSumFn.apply(gp, False).reshape(bias.shape).to(bias.dtype)
        if ns is not None and need[6]:
            gns = GF.PixelDotFn.apply(gp, noise).reshape(ns.shape).to(ns.dtype)
        gz = gp
        if dinv is not None:
            if need[3]:
                z = GF.Conv2dFn.apply(xs, weight, pad, pad)  # recomputed: the forward kernel stores only y
                gdinv = GF.ChanDotFn.apply(gp, z, True).reshape(dinv.shape).to(dinv.dtype)
            gz = GF.ChanScaleFn.apply(gp, dinv)
        if need[0] or need[2]:
            gxs = GF.Conv2dDgradFn.apply(gz, weight, pad, pad)
            if need[2]:
                gs = GF.ChanDotFn.apply(gxs, xu, True).reshape(s.shape).to(s.dtype)
            if need[0]:
                gx = GF.ChanScaleFn.apply(gxs, s) if s is not None else gxs
                if up2:  # adjoint of nearest x2: sum of each 2x2 block
                    gx = GF.UpFirDnFn.apply(gx, GF._ones_taps(gx, 1.0), 1, 2, 0, 0, 0, 0)
        if need[1]:
            gw = GF.Conv2dWgradFn.apply(xs, gz, kh, kw, pad, pad).to(weight.dtype)
        return gx, gw, gs, gdinv, gb, None, gns, None, None


class ModulatedConv2d(nn.Conv2d):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, z_dim: int, demodulation: bool = True) -> None:
        super().__init__(in_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.style_weight = nn.Linear(z_dim, in_channels)  # A in the paper
        self.demodulation = demodulation

    _styles = None  # (N, Cin) set by StyleGan2Generator.forward: every layer's A(w) + 1 from ONE GEMM
    _dinv = None    # (N, Cout) set there too when all layers' demodulation factors came from one launch

    def style_and_demod(self, w: Tensor):
        """s = A(w) + 1 (stylegan2.py:30) and the demodulation factor of F.normalize(dim=(2,3,4),
        eps=1e-8) (stylegan2.py:33), both in fp32, never forming the (N,Cout,Cin,k,k) tensor."""
        with torch.autocast("cuda", enabled=False):
            s, self._styles = self._styles, None
            if s is None:
                s = F.linear(w.float(), self.style_weight.weight, self.style_weight.bias) + 1  # (N, Cin)
            dinv, self._dinv = self._dinv, None
            if not self.demodulation:
                dinv = None
            elif dinv is None:
                dinv = GF.DemodFn.apply(s, self.weight)  # (N, Cout); raises on CPU tensors like every op here
        return s, dinv

    def fused(self, x: Tensor, w: Tensor, noise: Optional[Tensor], noise_scale: Optional[Tensor], slope: float, up2: bool):
        x = to_act(x)
        s, dinv = self.style_and_demod(w)
        return ModConvFn.apply(x, self.weight, s, dinv, self.bias, noise, noise_scale, slope, up2)

    def forward(self, x: Tensor, w: Tensor) -> Tensor:
        return self.fused(x, w, None, None, 1.0, False)


class StyleGan2GeneratorBlock(nn.Module):
    """blur_size (default None = the reference's block): a `Blur(blur_size)` FIR low-pass in the slot the reference marks
    `# TODO: blur` (stylegan2.py:47), between the upsampling modulated convolution and its Noise -- SURVEY.md 8 f1."""

    def __init__(self, in_dim: int, out_dim: int, z_dim: int, img_channels: int, first_block: bool = False,
                 blur_size: Optional[int] = None) -> None:
        super().__init__()
        self.layers = nn.ModuleList()
        if not first_block:
            self.layers.extend([Upsample(scale_factor=2.0), ModulatedConv2d(in_dim, out_dim, 3, z_dim)])
            if blur_size is not None:
                from .nvidia_ops import Blur

                self.layers.append(Blur(blur_size))
            self.layers.extend([Noise(out_dim), LeakyReLU(0.2)])
        self.layers.extend([ModulatedConv2d(out_dim, out_dim, 3, z_dim), Noise(out_dim), LeakyReLU(0.2)])
        self.to_rgb = ModulatedConv2d(out_dim, img_channels, 1, z_dim, demodulation=False)

    def forward(self, x: Tensor, w_embs: Tensor):
        """Walks self.layers like the reference (stylegan2.py:61-64) but hands each
        [Upsample?, ModulatedConv2d, Noise, LeakyReLU] group to ONE fused kernel launch."""
        layers = list(self.layers)
        i = 0
        x = to_act(x)
        while i < len(layers):
            up2 = False
            if isinstance(layers[i], Upsample):
                up2 = True
                i += 1
            conv = layers[i]
            if (
                isinstance(conv, ModulatedConv2d)
                and i + 2 < len(layers)
                and isinstance(layers[i + 1], Noise)
                and isinstance(layers[i + 2], LeakyReLU)
            ):
                noise_mod, act = layers[i + 1], layers[i + 2]
                n, _, h, w = x.shape
                ho, wo = (2 * h, 2 * w) if up2 else (h, w)
                # one draw per Noise forward, same shape/dtype/order as the reference (stylegan.py:26)
                draw = GF.noise_draw(n, ho, wo, x.device, x.dtype)
                x = conv.fused(x, w_embs, draw, noise_mod.scale, act.negative_slope, up2)
                i += 3
            elif (
                isinstance(conv, ModulatedConv2d)
                and i + 3 < len(layers)
                and type(layers[i + 1]).__name__ == "Blur"
                and isinstance(layers[i + 2], Noise)
                and isinstance(layers[i + 3], LeakyReLU)
            ):
                # blur-enabled block: [Upsample +] ModConv + bias in one launch, the FIR pass, then Noise + LeakyReLU in one
                blur, noise_mod, act = layers[i + 1], layers[i + 2], layers[i + 3]
                x = blur(conv.fused(x, w_embs, None, None, 1.0, up2))
                n, _, ho, wo = x.shape
                draw = GF.noise_draw(n, ho, wo, x.device, x.dtype)
                x = GF.affine_act(x, noise=draw, ns=noise_mod.scale, slope=act.negative_slope)
                i += 4
            else:  # unfused fallback for a re-arranged ModuleList: same ops, one launch each
                if up2:
                    x = GF.nearest_up2(x)
                x = conv(x, w_embs) if isinstance(conv, ModulatedConv2d) else conv(x)
                i += 1
        return x, self.to_rgb(x, w_embs)


class StyleGan2Generator(nn.Module):
    def __init__(self, img_size: int, img_channels: int = 3, z_dim: int = 128, base_dim: int = 16, blur_size: Optional[int] = None) -> None:
        super().__init__()
        self.mapping_network = nn.Sequential(nn.LayerNorm(z_dim))
        for _ in range(8):
            self.mapping_network.extend([nn.Linear(z_dim, z_dim), nn.LeakyReLU(0.2)])

        self.learned_input = nn.Parameter(torch.ones(1, 512, 4, 4))

        out_ch = min(base_dim * img_size // 4, 512)
        self.blocks = nn.ModuleList()
        self.blocks.append(StyleGan2GeneratorBlock(512, out_ch, z_dim, img_channels, first_block=True))
        in_ch = out_ch

        depth = int(math.log2(img_size // 4))
        for i in range(depth):
            out_ch = min(base_dim * img_size // 4 // 2 ** (i + 1), 512)
            self.blocks.append(StyleGan2GeneratorBlock(in_ch, out_ch, z_dim, img_channels, blur_size=blur_size))
            in_ch = out_ch

        self.reset_parameters()

    def reset_parameters(self):
        self.apply(init_weights)

    def _all_styles(self, w: Tensor) -> None:
        """Every ModulatedConv2d's style s = A_l(w) + 1 (stylegan2.py:30) from one GEMM over the stacked A_l:
        all layers see the same w (no style mixing, stylegan2.py:93-99), and 20 separate (N,128)x(128,Cin)
        products are 20 latency-bound launches forward and 40 backward."""
        mods = [m for m in self.modules() if isinstance(m, ModulatedConv2d)]
        with torch.autocast("cuda", enabled=False):
            a_all = torch.cat([m.style_weight.weight for m in mods], 0)
            b_all = torch.cat([m.style_weight.bias for m in mods], 0)
            s_all = F.linear(w.float(), a_all, b_all) + 1
            parts = GF.SplitColsFn.apply(s_all, tuple(m.in_channels for m in mods))  # one launch, not a copy per layer
            for m, s in zip(mods, parts):
                m._styles = s
                m._dinv = None
            # the demodulation factors of all layers from one launch too (else: DemodFn layer by layer, style_and_demod)
            dem, soffs, acc = [], [], 0
            for m in mods:
                if m.demodulation:
                    dem.append(m)
                    soffs.append(acc)
                acc += m.in_channels
            dinvs = GF.demod_all(s_all, soffs, [m.weight for m in dem])
            if dinvs is not None:
                for m, d in zip(dem, dinvs):
                    m._dinv = d

    def forward(self, z: Tensor) -> Tensor:
        # mapping network: LayerNorm + 8 x (128x128 Linear + LeakyReLU) on (N,128), fp32 -- one launch (mapping.hip)
        with torch.autocast("cuda", enabled=False):
            w = GF.mapping_network(self.mapping_network, z.float())
        self._all_styles(w)
        dt = GF.act_dtype(z)
        x = self.learned_input.to(dt).expand(z.shape[0], -1, -1, -1)
        if getattr(self, "bulk_noise", False) and z.is_cuda:
            # every Noise layer's draw as a slice of one launch (opt-in: the order of the random stream changes)
            per_sample = sum((4 * 2 ** i) ** 2 * sum(isinstance(m, Noise) for m in block.layers) for i, block in enumerate(self.blocks))
            with GF.bulk_noise(z.shape[0] * (per_sample + 8 * len(self.blocks) * 2), z.device, dt):
                return self._synthesis(x, w)
        return self._synthesis(x, w)

    def _synthesis(self, x: Tensor, w: Tensor) -> Tensor:
        x, y = self.blocks[0](x, w)
        for block in self.blocks[1:]:
            x, new_y = block(x, w)
            # RGB skip: y = upsample(y) + new_y (stylegan2.py:99) in one fused pass
            y = GF.up2_add(y, new_y)
        return y
