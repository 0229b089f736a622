"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Golden-vector generator.  Runs ONLY in the build container: imports the reference from
/root/reference on CPU (fp32), applies the harness-side patches listed in SURVEY.md section 8c
(never edits the reference), and writes inputs + expected outputs to tests/golden/*.pt.
The fixtures are data (tensors, floats); they load with torch.load(weights_only=True).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py

Harness-side patches:
  P1  StyleGan2GeneratorBlock.forward passes w_embs to to_rgb (shipped code omits it: stylegan2.py:64)
  P2  every nn.LeakyReLU inside the discriminator gets inplace=False (progressive_gan.py:44 + :54 break autograd)
  P3  base_dim chosen so that the first generator block is 512 channels (stylegan2.py:43-58,74)
"""
from __future__ import annotations

import copy
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import torch  # noqa: E402
from torch import nn  # noqa: E402

import diff_augment as ref_da  # noqa: E402  (reference)
import losses as ref_losses  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)
from modelling import build_discriminator, build_generator  # noqa: E402  (reference)
from modelling import nvidia_ops as ref_ops  # noqa: E402
from modelling import progressive_gan as ref_pg  # noqa: E402
from modelling import stylegan as ref_sg  # noqa: E402
from modelling import stylegan2 as ref_sg2  # noqa: E402

from oracle import gan_oracle as O  # noqa: E402
from oracle.detparams import det_state, det_uniform, projections, recipe_state  # noqa: E402
from oracle.tape import RandTape  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


# ---- P1 -------------------------------------------------------------------------------
def _block_forward(self, x, w_embs):
    for layer in self.layers:
        x = layer(x, w_embs) if isinstance(layer, ref_sg2.ModulatedConv2d) else layer(x)
    return x, self.to_rgb(x, w_embs)


ref_sg2.StyleGan2GeneratorBlock.forward = _block_forward


# ---- P2 -------------------------------------------------------------------------------
def out_of_place(model: nn.Module) -> nn.Module:
    for m in model.modules():
        if isinstance(m, nn.LeakyReLU):
            m.inplace = False
    return model


def save(name: str, obj) -> None:
    path = os.path.join(OUT, name + ".pt")
    torch.save(obj, path)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


def sd_of(m: nn.Module):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def grads_of(loss, tensors):
    gs = torch.autograd.grad(loss, tensors, allow_unused=True)
    return [None if g is None else g.detach().clone() for g in gs]


def named_grads(model: nn.Module):
    return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


# =========================================================================================
def gold_modconv():
    out = {}
    for tag, k, demod in (("k3_demod", 3, True), ("k1_plain", 1, False)):
        torch.manual_seed(11)
        m = ref_sg2.ModulatedConv2d(16, 32, k, 128, demodulation=demod)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * (0.3 if p.ndim > 1 else 0.2))
        x = torch.randn(4, 16, 8, 8, requires_grad=True)
        w = torch.randn(4, 128, requires_grad=True)
        params = dict(m.named_parameters())
        leaves = [x, w] + list(params.values())
        names = ["x", "w"] + list(params.keys())
        y = m(x, w)
        g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
        # double backward: differentiate |dL/dx|^2 + |dL/dw|^2 once more
        second = g1[0].square().sum() + g1[1].square().sum()
        g2 = torch.autograd.grad(second, leaves, allow_unused=True)
        out[tag] = {
            "state": sd_of(m),
            "x": x.detach().clone(),
            "w": w.detach().clone(),
            "y": y.detach().clone(),
            "grad": {n: g.detach().clone() for n, g in zip(names, g1)},
            "grad2": {n: (None if g is None else g.detach().clone()) for n, g in zip(names, g2)},
        }
    save("modconv", out)


def gold_upfirdn2d():
    cases = []
    torch.manual_seed(21)
    specs = []
    for ksz in range(2, 8):
        kern = ref_ops.Blur.make_kernel(ksz)
        p1 = (ksz - 1) // 2
        p2 = ksz - 1 - p1
        for up, down in ((1, 1), (2, 1), (1, 2)):
            specs.append((kern, up, down, p1, p2, p1, p2))
    r3 = torch.randn(3, 3)
    specs += [
        (r3, 1, 1, 1, 1, 1, 1),  # the microbench configuration (benchmark_upfirdn2d.py:60)
        (r3, 2, 1, 1, 2, 1, 2),
        (r3, 1, 2, 1, 2, 1, 2),
        (r3, 1, 1, 2, 0, 0, 2),
        (r3, 2, 2, 2, 1, 0, 3),
        (torch.randn(4, 2), 2, 1, 0, 1, 2, 1),  # non-square taps
        (torch.randn(4, 4), 1, 1, -1, 2, 2, -1),  # negative pad = crop
        (torch.randn(2, 2), 3, 2, 1, 1, 0, 2),
        (torch.ones(2, 2), 2, 1, 1, 0, 1, 0),  # nearest-up identity
        (torch.ones(2, 2) / 4, 1, 2, 0, 0, 0, 0),  # avg-pool identity
    ]
    for kern, up, down, px1, px2, py1, py2 in specs:
        x = torch.randn(2, 5, 8, 6, requires_grad=True)
        args = (kern, up, down, px1, px2, py1, py2)
        y_plain = ref_ops.upfirdn2d(x, *args)
        (g1,) = torch.autograd.grad(y_plain.square().sum(), x, create_graph=True)
        (g2,) = torch.autograd.grad(g1.square().sum(), x)
        # The Function's backward (argument swap, nvidia_ops.py:53) only yields the right shape when
        # p1 + p2 == k - 1 and the sizes divide; where it runs it must equal plain autograd.
        fn_ok = True
        try:
            y = ref_ops.UpFIRDn2d.apply(x, *args)
            assert torch.equal(y, y_plain)
            (f1,) = torch.autograd.grad(y.square().sum(), x, create_graph=True)
            (f2,) = torch.autograd.grad(f1.square().sum(), x)
            assert torch.allclose(f1, g1, atol=1e-5) and torch.allclose(f2, g2, atol=1e-3, rtol=1e-4)
        except RuntimeError:
            fn_ok = False
        y = y_plain
        cases.append(
            {
                "kernel": kern.clone(),
                "args": [up, down, px1, px2, py1, py2],
                "x": x.detach().clone(),
                "y": y.detach().clone(),
                "g1": g1.detach().clone(),
                "g2": g2.detach().clone(),
                "function_backward_runs": fn_ok,
            }
        )
    ident = {}
    x = torch.randn(2, 3, 6, 4)
    ident["x"] = x
    ident["nearest_up2"] = nn.Upsample(scale_factor=2.0)(x)
    ident["avg_pool2"] = nn.AvgPool2d(2)(x)
    blur = {}
    xb = torch.randn(2, 4, 8, 8)
    blur["x"] = xb
    for ksz, up, down in ((3, 1, 1), (4, 2, 1), (4, 1, 2), (2, 1, 1), (7, 1, 1)):
        blur[f"k{ksz}_u{up}_d{down}"] = ref_ops.Blur(ksz, up, down)(xb)
    save(
        "upfirdn2d",
        {
            "cases": cases,
            "make_kernel": {str(k): ref_ops.Blur.make_kernel(k) for k in range(2, 8)},
            "identities": ident,
            "blur": blur,
        },
    )


def gold_small_ops():
    torch.manual_seed(31)
    out = {}
    # Noise with recorded draw (stylegan.py:19-27)
    noise_mod = ref_sg.Noise(6)
    with torch.no_grad():
        noise_mod.scale.copy_(torch.randn(6, 1, 1))
    x = torch.randn(3, 6, 4, 4)
    with RandTape() as tape:
        y = noise_mod(x)
    out["noise"] = {"x": x, "scale": noise_mod.scale.detach().clone(), "draw": tape.tape[0], "y": y.detach()}
    # MinibatchStdDev fwd / grad / double-grad (progressive_gan.py:20-36)
    x = torch.randn(8, 5, 4, 4, requires_grad=True)
    y = ref_pg.MinibatchStdDev()(x)
    wgt = torch.randn_like(y)
    (g1,) = torch.autograd.grad((y * wgt).sum() + y[:, -1].square().sum(), x, create_graph=True)
    (g2,) = torch.autograd.grad(g1.square().sum(), x)
    out["mbstd"] = {"x": x.detach().clone(), "y": y.detach().clone(), "wgt": wgt, "g1": g1.detach().clone(), "g2": g2.detach().clone()}
    # fused bias + leaky-relu: composition oracle
    x = torch.randn(2, 6, 5, 5, requires_grad=True)
    b = torch.randn(6, requires_grad=True)
    y = nn.LeakyReLU(0.2)(x + b.view(1, -1, 1, 1))
    gx, gb = torch.autograd.grad(y.square().sum(), [x, b], create_graph=True)
    ggx, ggb = torch.autograd.grad(gx.square().sum() + gb.square().sum(), [x, b])
    out["bias_lrelu"] = {
        "x": x.detach().clone(),
        "b": b.detach().clone(),
        "y": y.detach().clone(),
        "gx": gx.detach().clone(),
        "gb": gb.detach().clone(),
        "ggx": ggx.detach().clone(),
        "ggb": ggb.detach().clone(),
    }
    # the five losses on fixed logits (losses.py:27-83)
    d_r = torch.randn(8)
    d_f = torch.randn(8)
    lo = {"d_reals": d_r, "d_fakes": d_f}
    for name in ("gan", "lsgan", "wgan", "hinge", "relativistic-gan"):
        crit = ref_losses.get_loss(name)
        lo[name] = [float(crit.d_loss(d_r, d_f)), float(crit.g_loss(d_f, lambda: d_r))]
    out["losses"] = lo
    # EMA after 0, 99, 100, 101, 103 steps (utils.py:75-99)
    lin = nn.Linear(4, 3)
    ema = ref_utils.EMA(lin, beta=0.999, warmup=100)
    snaps = {"init": sd_of(lin)}
    drift = torch.randn(3, 4) * 0.01
    for step in range(1, 104):
        with torch.no_grad():
            lin.weight.add_(drift)
            lin.bias.add_(0.001 * step)
        ema.step()
        if step in (1, 99, 100, 101, 103):
            snaps[str(step)] = {"model": sd_of(lin), "ema": sd_of(ema.ema_model), "counter": int(ema.counter)}
    snaps["drift"] = drift
    out["ema"] = snaps
    save("small_ops", out)


def gold_disc():
    torch.manual_seed(41)
    out = {}
    # one block (progressive_gan.py:40-54)
    blk = out_of_place(ref_pg.ProgressiveGanDiscriminatorBlock(16, 32))
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    x = torch.randn(4, 16, 8, 8, requires_grad=True)
    y = blk(x)
    leaves = [x] + list(blk.parameters())
    names = ["x"] + [k for k, _ in blk.named_parameters()]
    g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
    g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
    out["block"] = {
        "state": sd_of(blk),
        "x": x.detach().clone(),
        "y": y.detach().clone(),
        "grad": {n: g.detach().clone() for n, g in zip(names, g1)},
        "grad2": {n: (None if g is None else g.detach().clone()) for n, g in zip(names, g2)},
    }
    # tiny full discriminator: hinge, R1, WGAN-GP (losses.py:64-69,89-103)
    disc = out_of_place(build_discriminator("stylegan2", img_size=16, base_dim=8))
    with torch.no_grad():
        for k, p in disc.named_parameters():
            if k.endswith("scale"):
                p.copy_(0.5 + torch.rand_like(p))  # 1e-4 would hide the residual branch
            elif k.endswith("bias"):
                p.copy_(torch.randn_like(p) * 0.1)
    reals = torch.randn(8, 3, 16, 16)
    fakes = torch.randn(8, 3, 16, 16)
    rec = {"state": sd_of(disc), "reals": reals.clone(), "fakes": fakes.clone()}
    rec["logits_reals"] = disc(reals).detach().clone()

    disc.zero_grad()
    hinge = ref_losses.HingeLoss.d_loss(disc(reals), disc(fakes))
    hinge.backward()
    rec["hinge"] = {"value": float(hinge), "grads": named_grads(disc)}

    disc.zero_grad()
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    r1 = ref_losses.r1_regularizer(disc, r, fakes, d_r)
    r1.backward()
    rec["r1"] = {"value": float(r1), "grads": named_grads(disc), "grad_reals": r.grad.detach().clone()}

    disc.zero_grad()
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    d_f = disc(fakes)
    total = ref_losses.HingeLoss.d_loss(d_r, d_f) + ref_losses.r1_regularizer(disc, r, fakes, d_r)
    total.backward()
    rec["hinge_r1"] = {"value": float(total), "grads": named_grads(disc)}

    disc.zero_grad()
    with RandTape() as tape:
        gp = ref_losses.wgan_gp_regularizer(disc, reals, fakes, None)
    gp.backward()
    rec["wgan_gp"] = {"value": float(gp), "alpha": tape.tape[0], "grads": named_grads(disc)}
    out["disc16"] = rec
    save("disc", out)


def gold_generator():
    """StyleGan2Generator(16, base_dim=128): smallest size whose first block is 512-ch (P3).
    4.6 M parameters -> built from oracle.detparams (seed 5), big grads stored as projections."""
    gen = build_generator("stylegan2", img_size=16, base_dim=128)
    shapes = {k: tuple(v.shape) for k, v in gen.state_dict().items()}
    assert shapes == O.stylegan2_generator_shapes(16, base_dim=128), "shape table drifted from the reference"
    gen.load_state_dict(det_state(shapes, seed=5))
    z = det_uniform((4, 128), 901) * 3.0
    with RandTape() as tape:
        img = gen(z)
    loss = (img * det_uniform(tuple(img.shape), 902)).sum() + img.square().mean()
    loss.backward()
    grads = {}
    for k, p in gen.named_parameters():
        grads[k] = p.grad.detach().clone() if p.numel() <= 4096 else projections(p.grad)
    save(
        "generator16",
        {"seed": 5, "img_size": 16, "base_dim": 128, "z": z, "noise": tape.tape, "img": img.detach().clone(), "loss": float(loss), "grads": grads},
    )


def gold_dcgan():
    torch.manual_seed(61)
    disc = build_discriminator("dcgan", img_size=64, base_dim=8, norm="none")
    gen = build_generator("dcgan", img_size=64, base_dim=8)
    with torch.no_grad():
        for net in (disc, gen):
            for p in net.parameters():
                if p.ndim > 1:
                    p.mul_(5.0)  # N(0, 0.02) leaves everything ~0; widen so grads are informative
                else:
                    p.add_(torch.randn_like(p) * 0.1)
    z = torch.randn(4, 128)
    reals = torch.randn(4, 3, 64, 64)
    fakes = gen(z)
    rec = {"d_state": sd_of(disc), "g_state": sd_of(gen), "z": z, "reals": reals, "fakes": fakes.detach().clone()}
    rec["logits_reals"] = disc(reals).detach().clone()
    disc.zero_grad()
    with RandTape() as tape:
        gp = ref_losses.wgan_gp_regularizer(disc, reals, fakes.detach(), None)
    loss = ref_losses.WganLoss.d_loss(disc(reals), disc(fakes.detach())) + gp
    loss.backward()
    rec["wgan_gp"] = {"value": float(gp), "loss": float(loss), "alpha": tape.tape[0], "grads": named_grads(disc)}
    gen.zero_grad()
    gl = ref_losses.WganLoss.g_loss(disc(gen(z)), None)
    gl.backward()
    rec["g_loss"] = {"value": float(gl), "grads": named_grads(gen)}
    save("dcgan", rec)


def gold_trajectory():
    """10 generator iterations of the train_celeba.py:111-149 loop body (hinge + R1, n_disc=1, grad_accum=1,
    fp32, no autocast) on reference modules/losses/EMA; 16x16, batch 4, Adam(2e-4, betas (0.0, 0.9)).
    The loop body is restated here from the reference by reading (train_celeba.py itself needs wandb /
    torchvision / CelebA download and cannot be imported)."""
    gen = build_generator("stylegan2", img_size=16, base_dim=128)
    disc = out_of_place(build_discriminator("stylegan2", img_size=16, base_dim=16))
    g_shapes = {k: tuple(v.shape) for k, v in gen.state_dict().items()}
    d_shapes = {k: tuple(v.shape) for k, v in disc.state_dict().items()}
    assert d_shapes == O.stylegan2_discriminator_shapes(16, base_dim=16)
    gen.load_state_dict(det_state(g_shapes, seed=9, bias_amp=0.0, scale_amp=0.05))
    disc.load_state_dict(det_state(d_shapes, seed=10, bias_amp=0.0, scale_amp=0.5))
    ema = ref_utils.EMA(gen, warmup=3)
    optim_d = torch.optim.Adam(disc.parameters(), 2e-4, betas=(0.0, 0.9))
    optim_g = torch.optim.Adam(gen.parameters(), 2e-4, betas=(0.0, 0.9))
    criterion = ref_losses.get_loss("hinge")
    regularizer = ref_losses.get_regularizer("r1")
    steps = []
    reals_all = []
    tape_all = []
    for step in range(10):
        reals = det_uniform((4, 3, 16, 16), 1000 + step) * 2.0
        reals_all.append(reals.clone())
        with RandTape() as tape:
            cached = reals.clone()
            reals.requires_grad_()
            zs = torch.randn(reals.shape[0], 128)
            with torch.no_grad():
                fakes = gen(zs)
            d_reals = disc(reals)
            d_fakes = disc(fakes)
            loss_d = criterion.d_loss(d_reals, d_fakes)
            loss_d = loss_d + regularizer(disc, reals, fakes, d_reals)
            loss_d.backward()
            optim_d.step()
            optim_d.zero_grad()
            disc.requires_grad_(False)
            zs = torch.randn(reals.shape[0], 128)
            loss_g = criterion.g_loss(disc(gen(zs)), lambda: disc(cached))
            loss_g.backward()
            optim_g.step()
            optim_g.zero_grad()
            disc.requires_grad_(True)
            ema.step()
        tape_all.append(tape.tape)
        steps.append([float(loss_d), float(loss_g), float(d_reals.mean()), float(d_fakes.mean())])
    final = {
        "gen": {k: projections(v) for k, v in gen.state_dict().items()},
        "disc": {k: projections(v) for k, v in disc.state_dict().items()},
        "ema": {k: projections(v) for k, v in ema.ema_model.state_dict().items()},
    }
    save(
        "trajectory16",
        {"gen_seed": 9, "disc_seed": 10, "ema_warmup": 3, "reals_salt0": 1000, "tape": tape_all, "steps": torch.tensor(steps, dtype=torch.float64), "final": final},
    )


def _ref_train_iteration(gen, disc, optim_g, optim_d, ema, criterion, regularizer, next_reals, n_disc, z_dim, clip):
    """One `while` body of train_celeba.py:111-149 with grad_accum = 1, restated from the reference by reading (the
    script itself needs wandb / torchvision / a CelebA download), on the reference's modules, losses and EMA."""
    d_losses, cached = [], None
    for i in range(n_disc):
        if clip:  # train_celeba.py:115-118
            with torch.no_grad():
                for p in disc.parameters():
                    p.clip_(-0.01, 0.01)
        reals = next_reals()
        if i == n_disc - 1:
            cached = reals.clone()
        reals.requires_grad_()
        zs = torch.randn(reals.shape[0], z_dim)
        with torch.no_grad():
            fakes = gen(zs)
        d_reals = disc(reals)
        d_fakes = disc(fakes)
        loss_d = criterion.d_loss(d_reals, d_fakes)
        if regularizer is not None:
            loss_d = loss_d + regularizer(disc, reals, fakes, d_reals)
        loss_d.backward()
        optim_d.step()
        optim_d.zero_grad()
        d_losses.append(float(loss_d))
    disc.requires_grad_(False)
    zs = torch.randn(reals.shape[0], z_dim)
    loss_g = criterion.g_loss(disc(gen(zs)), lambda: disc(cached))
    loss_g.backward()
    optim_g.step()
    optim_g.zero_grad()
    disc.requires_grad_(True)
    ema.step()
    return d_losses, float(loss_g)


def gold_trainer_recipes():
    """The train loop on the README recipes the 16 x 16 trajectory does not reach (README.md:55,60):
      dcgan_wgan_gp  DCGAN, WGAN + gradient penalty, n_disc = 5, Adam(1e-4, (0, 0.9)), 64 x 64, batch 8, 3 iterations
      dcgan_wgan_clip  DCGAN, WGAN with weight clipping (no regularizer: train_celeba.py:115-118), n_disc = 2, 2 iterations
      stylegan2_64   StyleGAN2 64 x 64, base_dim 64 both networks, hinge + R1, Adam(2e-4, (0, 0.9)), batch 8, 1 iteration
    Deterministic parameters (oracle.detparams: only shape tables and seeds are stored), deterministic reals, the random
    draws recorded; per iteration every D loss and the G loss, at the end a fingerprint of every parameter."""
    out = {}
    # (wgain: the DCGAN nets get a quarter of det_state's kaiming-variance weights -- at full width the WGAN losses grow
    # 5x per iteration and the third iteration amplifies fp32 rounding past 1e-4; the reference's own init is N(0, 0.02))
    for name, model, size, kw_g, kw_d, method, reg, n_disc, lr, iters, batch, wgain in (
        ("dcgan_wgan_gp", "dcgan", 64, {}, {"norm": "none"}, "wgan", "wgan-gp", 5, 1e-4, 3, 8, 0.25),
        ("dcgan_wgan_clip", "dcgan", 64, {}, {"norm": "none"}, "wgan", "none", 2, 1e-4, 2, 8, 0.25),
        ("stylegan2_64", "stylegan2", 64, {"base_dim": 64}, {"base_dim": 64}, "hinge", "r1", 1, 2e-4, 1, 8, 1.0),
    ):
        gen = build_generator(model, img_size=size, **kw_g)
        disc = out_of_place(build_discriminator(model, img_size=size, **kw_d))
        g_shapes = {k: tuple(v.shape) for k, v in gen.state_dict().items()}
        d_shapes = {k: tuple(v.shape) for k, v in disc.state_dict().items()}
        gseed, dseed = 21, 22
        gen.load_state_dict(recipe_state(g_shapes, gseed, 0.05, wgain))
        disc.load_state_dict(recipe_state(d_shapes, dseed, 0.5, wgain))
        ema = ref_utils.EMA(gen, warmup=2)
        optim_d = torch.optim.Adam(disc.parameters(), lr, betas=(0.0, 0.9))
        optim_g = torch.optim.Adam(gen.parameters(), lr, betas=(0.0, 0.9))
        criterion = ref_losses.get_loss(method)
        regularizer = ref_losses.get_regularizer(reg)
        salt = [5000]

        def next_reals():
            salt[0] += 1
            return det_uniform((batch, 3, size, size), salt[0]) * 2.0

        steps, tapes = [], []
        for _ in range(iters):
            with RandTape() as tape:
                d_losses, g_loss = _ref_train_iteration(gen, disc, optim_g, optim_d, ema, criterion, regularizer, next_reals, n_disc, 128,
                                                        clip=(method == "wgan" and reg == "none"))
            tapes.append(tape.tape)
            steps.append(d_losses + [g_loss])
        out[name] = {
            "model": model, "img_size": size, "gen_kwargs": kw_g, "disc_kwargs": kw_d, "method": method, "regularizer": reg, "n_disc": n_disc,
            "lr": lr, "batch": batch, "wgain": wgain, "gen_seed": gseed, "disc_seed": dseed, "gen_shapes": {k: list(v) for k, v in g_shapes.items()},
            "disc_shapes": {k: list(v) for k, v in d_shapes.items()}, "reals_salt0": 5000, "ema_warmup": 2, "tape": tapes,
            "steps": torch.tensor(steps, dtype=torch.float64),
            "final": {"gen": {k: projections(v) for k, v in gen.state_dict().items()}, "disc": {k: projections(v) for k, v in disc.state_dict().items()},
                      "ema": {k: projections(v) for k, v in ema.ema_model.state_dict().items()}},
        }
        print(name, steps)
    save("trainer_recipes", out)


def gold_generators_v1():
    """StyleGanGenerator and ProgressiveGanGenerator (stylegan.py:50-85, progressive_gan.py:103-133) at 16x16, base_dim 16:
    deterministic parameters (oracle.detparams, only the shape table is stored), recorded noise, image, loss and every
    parameter gradient (fingerprints for the big tensors)."""
    out = {}
    for name, seed in (("stylegan", 31), ("progressive_gan", 32)):
        gen = build_generator(name, img_size=16, base_dim=16)
        for m in gen.modules():
            if isinstance(m, nn.LeakyReLU):
                m.inplace = False
        shapes = {k: tuple(v.shape) for k, v in gen.state_dict().items()}
        gen.load_state_dict(det_state(shapes, seed=seed, bias_amp=0.1, scale_amp=0.1))
        z = det_uniform((4, 128), 940 + seed) * 3.0
        with RandTape() as tape:
            img = gen(z)
        wgt = det_uniform(tuple(img.shape), 950 + seed)
        loss = (img * wgt).sum() + img.square().mean()
        loss.backward()
        grads = {k: (p.grad.detach().clone() if p.numel() <= 4096 else projections(p.grad)) for k, p in gen.named_parameters()}
        out[name] = {"seed": seed, "shapes": {k: list(v) for k, v in shapes.items()}, "z": z, "noise": tape.tape, "img": img.detach().clone(),
                     "loss": float(loss), "grads": grads}
    save("generators_v1", out)


def gold_resample_convs():
    """up_conv_blur / blur_conv_down (nvidia_ops.py:81-123), all three branches of each (blur_size None, kernel_size 1,
    kernel_size 3), blur sizes 3 and 4: forward, first-order gradients of every leaf, second order through x."""
    torch.manual_seed(71)
    out = {}
    for fn_name, fn in (("up_conv_blur", ref_ops.up_conv_blur), ("blur_conv_down", ref_ops.blur_conv_down)):
        for ksz, blur in ((3, None), (1, 3), (3, 3), (3, 4), (1, 4)):
            m = fn(6, 10, ksz, blur)
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.randn_like(p) * 0.3)
            x = torch.randn(2, 6, 8, 8, requires_grad=True)
            y = m(x)
            names = ["x"] + [k for k, _ in m.named_parameters()]
            leaves = [x] + list(m.parameters())
            g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
            g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
            out[f"{fn_name}_k{ksz}_b{blur}"] = {
                "fn": fn_name, "kernel_size": ksz, "blur_size": blur,
                "state": sd_of(m), "x": x.detach().clone(), "y": y.detach().clone(),
                "grad": {n: g.detach().clone() for n, g in zip(names, g1)},
                "grad2": {n: (None if g is None else g.detach().clone()) for n, g in zip(names, g2)},
            }
    save("resample_convs", out)


def gold_diff_augment():
    """DiffAugment (diff_augment.py:54-60) and its pieces on CPU: the draws the reference makes from the global generator
    are recovered by re-seeding and drawing in the same order (:14, :19, :24 -> torch.rand(N, 1, 1, 1); :42 -> x then y;
    :48-49 -> y then x).  Output, gradient of a weighted + quadratic loss in x, and the gradient of that gradient's norm."""
    out = {}
    for tag, shape, seed in (("rgb16", (4, 3, 16, 16), 101), ("rgb24x40", (3, 3, 24, 40), 102), ("gray8", (2, 1, 8, 8), 103), ("rgb32", (2, 3, 32, 32), 104)):
        n, c, h, w = shape
        g = torch.Generator().manual_seed(seed)
        x0 = torch.randn(shape, generator=g)
        wgt = torch.randn(shape, generator=g)
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(seed)
        y = ref_da.DiffAugment()(x)
        torch.manual_seed(seed)
        b, s, k = (torch.rand(n, 1, 1, 1) for _ in range(3))
        tx, ty = ref_da.rand_int(-(w // 8), w // 8 + 1), ref_da.rand_int(-(h // 8), h // 8 + 1)
        oy, ox = ref_da.rand_int(0, h // 2), ref_da.rand_int(0, w // 2)
        loss = (y * wgt).sum() + y.square().mean()
        (g1,) = torch.autograd.grad(loss, x, create_graph=True)
        (g2,) = torch.autograd.grad(g1.square().sum(), x)
        assert torch.allclose(O.diff_augment(x0, b, s, k, tx, ty, oy, ox), y.detach(), atol=1e-6), tag
        pieces = {}
        for name, fn in () if tag != "rgb16" else (("brightness", ref_da.rand_brightness), ("saturation", ref_da.rand_saturation), ("contrast", ref_da.rand_contrast)):
            torch.manual_seed(seed + 7)
            pieces[name] = {"y": fn(x0.clone()), "u": None}
            torch.manual_seed(seed + 7)
            pieces[name]["u"] = torch.rand(n, 1, 1, 1)
        if tag == "rgb16":
            pieces["translate"] = {"y": ref_da.translate(x0.clone(), tx, ty)}
        out[tag] = {
            "x": x0, "wgt": wgt, "brightness": b, "saturation": s, "contrast": k, "translate": (tx, ty), "cutout": (oy, ox),
            "y": y.detach().clone(), "loss": float(loss.detach()), "grad": g1.detach().clone(), "grad2": g2.detach().clone(), "pieces": pieces,
        }
    save("diff_augment", out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:  # regenerate only the named fixtures: python oracle/make_goldens.py resample_convs
        for name in sys.argv[1:]:
            globals()["gold_" + name]()
        sys.exit(0)
    gold_modconv()
    gold_upfirdn2d()
    gold_small_ops()
    gold_disc()
    gold_generator()
    gold_dcgan()
    gold_trajectory()
    gold_resample_convs()
    gold_generators_v1()
    gold_diff_augment()
    gold_trainer_recipes()
