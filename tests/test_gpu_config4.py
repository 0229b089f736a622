"""GPU parity at the sizes and on the kernel paths that bench.py actually runs (BASELINE config 4: StyleGAN2
hinge + R1, 256x256, base_dim 16, bf16) -- every check against the CPU oracle (fp64 where a tolerance of 1e-5
needs it) or against reference-generated goldens:

  * upfirdn2d bf16 with more tiles than workgroups: the persistent multi-tile walk + register prefetch of
    upfirdn2d_mfma_k (the microbenchmark's path);
  * discriminator and generator at 256x256 / base_dim 16, batch 4, fp32 and bf16: outputs, hinge + R1 loss, every
    parameter gradient (the 16..64-channel thin kernels at 128x128 / 256x256 only exist at this size);
  * one bf16 train iteration at 256x256 and at 32x32 against the oracle's iteration;
  * the EMA module's 0 / 99 / 100 / 101 schedule, and the EMA forward after an update;
  * up_conv_blur / blur_conv_down, all branches, against the reference's outputs.

Tolerances: fp32 1e-5 forward / 1e-4 gradients that pass through a second-order graph; bf16 1e-2 per op
(north_star).  For whole-network bf16 quantities (six blocks deep, bf16 activations between all of them) the
per-tensor bound is stated at each assert."""
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL, TOL2, TOLBF = 1e-5, 1e-4, 1e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gan_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------------
# (N, C, H, W): tiles are 32 wide x (16 / channel-groups) tall x 16..64 channels; the grid is at most 1024
# workgroups, so each of these makes a workgroup walk >= 2 tiles (the last one an uneven split with ragged edges)
@pytest.mark.parametrize("shape", [(16, 64, 128, 128), (4, 16, 512, 512), (9, 32, 200, 136)])
def test_upfirdn2d_persistent_tiles(dev, shape):
    from gan_amd.modelling.nvidia_ops import Blur, upfirdn2d
    from oracle import gan_oracle as O

    torch.manual_seed(5)
    x = torch.randn(shape).bfloat16()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    k3 = torch.randn(3, 3)
    k4 = Blur.make_kernel(4)
    for kern, up, down, pads in [(k3, 1, 1, (1, 1, 1, 1)), (k4, 2, 1, (2, 1, 2, 1)), (k4, 1, 2, (1, 2, 1, 2))]:
        ref = O.upfirdn2d(x.float(), kern, up, down, *pads)  # fp32 tap loop on the CPU (bf16 inputs are exact in fp32)
        got = upfirdn2d(xd, kern.to(dev), up, down, *pads)
        assert got.dtype == torch.bfloat16 and got.shape == ref.shape
        # every element, not just the norm: a stale LDS row or a mixed-up prefetch is a local error
        diff = (got.float().cpu() - ref).abs()
        bound = 2.0 ** -7 * ref.abs() + 1e-2 * float(ref.abs().mean())  # bf16 rounding of the output + fp32 summation order
        assert bool((diff <= bound).all()), (shape, up, down, float(diff.max()))
        assert rel_err(got, ref) < TOLBF
        del ref, got, diff, bound
    # first-order gradient: the same kernel with flipped taps, shapes swapped
    xg = xd.clone().requires_grad_(True)
    y = upfirdn2d(xg, k3.to(dev), 1, 1, 1, 1, 1, 1)
    wgt = torch.randn(y.shape, device=dev, dtype=y.dtype).contiguous(memory_format=torch.channels_last)
    (g,) = torch.autograd.grad((y * wgt).sum(), xg)
    ref = O.upfirdn2d(wgt.float().cpu(), k3.flip(0, 1), 1, 1, 1, 1, 1, 1)
    assert rel_err(g, ref) < TOLBF


# ------------------------------------------------------------------------------------------------------
def _d_state(size=256, base=16, seed=21):
    from oracle import gan_oracle as O
    from oracle.detparams import det_state

    return det_state(O.stylegan2_discriminator_shapes(size, base_dim=base), seed=seed, bias_amp=0.1, scale_amp=0.5)


def _g_state(size=256, base=16, seed=22):
    from oracle import gan_oracle as O
    from oracle.detparams import det_state

    return det_state(O.stylegan2_generator_shapes(size, base_dim=base), seed=seed, bias_amp=0.1, scale_amp=0.05)


def _tensor_errs(got_grads, ref_grads):
    """{name: rel-err} for the tensors whose reference norm is not negligible next to the largest one (the others --
    e.g. bias gradients of the R1 term, which are non-zero only through MinibatchStdDev's coupling, ~1e-6 -- are
    checked in absolute terms by the caller), plus the rel-err of all gradients taken as one vector."""
    top = max(float(r.norm()) for r in ref_grads.values())
    errs, num, den = {}, 0.0, 0.0
    for k, ref in ref_grads.items():
        got = got_grads[k]
        got = torch.zeros_like(ref) if got is None else got.detach().double().cpu()
        if float(ref.norm()) >= 1e-4 * top:
            errs[k] = rel_err(got, ref)
        else:
            assert float((got - ref).norm()) <= 1e-4 * top, k
        num += float((got - ref).square().sum())
        den += float(ref.square().sum())
    return errs, (num / den) ** 0.5


def _oracle_d(state, reals, fakes, dtype, loss_kind, wa, wb, autocast=False):
    """Oracle discriminator loss + every parameter gradient.  loss_kind "hinge": hinge + R1 (what the step runs);
    "smooth": mean(d_r * wa) + mean(d_f * wb) + R1 with positive weights -- no hinge gates, and reals / fakes
    gradients add up instead of cancelling (synthetic reals and fakes are statistically alike: with d_f - d_r the
    gradient is a small difference of two large vectors and ANY bf16 path, the reference's included, is 20 % off)."""
    from oracle import gan_oracle as O

    sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in state.items()}
    r = reals.detach().clone().to(dtype).requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        d_r = O.discriminator(sd, r)
        d_f = O.discriminator(sd, fakes.to(dtype))
        if loss_kind == "hinge":
            loss = O.d_loss("hinge", d_r.float() if autocast else d_r, d_f.float() if autocast else d_f)
        else:
            loss = (d_r.float() * wa).mean() + (d_f.float() * wb).mean() if autocast else (d_r * wa.to(dtype)).mean() + (d_f * wb.to(dtype)).mean()
        loss = loss + O.r1_penalty(None, r, None, d_r.float() if autocast else d_r)
    loss.backward()
    grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad).double() for k, v in sd.items()}
    return d_r.detach(), d_f.detach(), float(loss.detach()), grads


def _hip_d(dev, state, reals, fakes, mixed, loss_kind, wa, wb):
    from gan_amd.losses import get_loss, get_regularizer
    from gan_amd.modelling import build_discriminator

    disc = build_discriminator("stylegan2", img_size=reals.shape[-1], base_dim=state["0.weight"].shape[0])
    disc.load_state_dict(state)
    disc = disc.to(dev)
    rg = reals.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
        d_r = disc(rg)
        d_f = disc(fakes.to(dev))
        if loss_kind == "hinge":
            loss = get_loss("hinge").d_loss(d_r, d_f)
        else:
            loss = (d_r * wa.to(dev)).mean() + (d_f * wb.to(dev)).mean()
        loss = loss + get_regularizer("r1")(disc, rg, fakes.to(dev), d_r)
    loss.backward()
    return d_r.detach(), d_f.detach(), float(loss.detach()), {k: p.grad for k, p in disc.named_parameters()}


def test_discriminator256_fp32_vs_oracle(dev):
    """ProgressiveGanDiscriminator(256, base_dim 16) -- bench.py's D -- batch 4, fp32: logits, loss value and the
    gradient of every parameter (first- and second-order paths through every D-block kernel) against the oracle in
    fp64, for hinge + R1 (the step's loss) and for a gate-free loss + R1."""
    from oracle.detparams import det_uniform

    state = _d_state()
    reals = det_uniform((4, 3, 256, 256), 311) * 2.0
    fakes = det_uniform((4, 3, 256, 256), 312) * 2.0
    wa, wb = det_uniform((4,), 5) + 1.0, det_uniform((4,), 6) + 1.0
    for kind in ("hinge", "smooth"):
        d_r, d_f, loss_ref, ref = _oracle_d(state, reals, fakes, torch.float64, kind, wa, wb)
        g_r, g_f, loss, got = _hip_d(dev, state, reals, fakes, False, kind, wa, wb)
        errs, total = _tensor_errs(got, ref)
        worst = max(errs.items(), key=lambda kv: kv[1])
        print(f"D256 fp32 {kind}: logits {rel_err(g_r, d_r):.2e} / {rel_err(g_f, d_f):.2e}, loss {loss:.6f} vs {loss_ref:.6f}, "
              f"all grads {total:.2e}, worst tensor {worst[0]} {worst[1]:.2e}")
        assert rel_err(g_r, d_r) < TOL and rel_err(g_f, d_f) < TOL
        assert math.isclose(loss, loss_ref, rel_tol=1e-5)
        assert total < TOL2, total
        # Per tensor 1e-4 -- except under the hinge loss, where the reals' and fakes' gradients nearly cancel (the
        # synthetic images are statistically alike) and one leaky-ReLU input within 1e-7 of zero taking the other
        # branch moves a 64-channel layer's gradient by 3e-4 of its (small) norm: 1e-3 there.
        assert worst[1] < (1e-3 if kind == "hinge" else TOL2), worst


def test_discriminator256_bf16_vs_oracle(dev):
    """The same discriminator under bf16 autocast (thin direct-convolution kernels at 256 / 128 / 64 pixels, LDS-DMA
    wide kernels below: these kernel families only exist in bf16) against the oracle in fp64.  The yardstick is the
    reference arithmetic's own bf16 error: the ORACLE under torch.autocast(cpu, bf16) against its fp64 run.  The HIP
    path must be within 1e-2 (north_star's per-op figure) or 1.5x that yardstick, whichever is larger."""
    from oracle.detparams import det_uniform

    state = _d_state()
    reals = det_uniform((4, 3, 256, 256), 311) * 2.0
    fakes = det_uniform((4, 3, 256, 256), 312) * 2.0
    wa, wb = det_uniform((4,), 5) + 1.0, det_uniform((4,), 6) + 1.0
    d_r, d_f, loss_ref, ref = _oracle_d(state, reals, fakes, torch.float64, "smooth", wa, wb)
    c_r, c_f, loss_cpu, cpu = _oracle_d(state, reals, fakes, torch.float32, "smooth", wa, wb, autocast=True)
    g_r, g_f, loss, got = _hip_d(dev, state, reals, fakes, True, "smooth", wa, wb)
    errs, total = _tensor_errs(got, ref)
    errs_cpu, total_cpu = _tensor_errs(cpu, ref)
    worst = max(errs, key=lambda k: errs[k] / max(TOLBF, errs_cpu[k]))
    print(f"D256 bf16: logits {rel_err(g_r, d_r):.2e} (CPU autocast {rel_err(c_r, d_r):.2e}), loss {loss:.5f} vs {loss_ref:.5f} "
          f"(CPU autocast {loss_cpu:.5f}), all grads {total:.2e} (CPU autocast {total_cpu:.2e}), "
          f"worst tensor {worst} {errs[worst]:.2e} (CPU autocast {errs_cpu[worst]:.2e})")
    assert rel_err(g_r, d_r) < max(TOLBF, 1.5 * rel_err(c_r, d_r))
    assert rel_err(g_f, d_f) < max(TOLBF, 1.5 * rel_err(c_f, d_f))
    assert abs(loss - loss_ref) <= max(TOLBF * abs(loss_ref), 1.5 * abs(loss_cpu - loss_ref))
    assert total < max(TOLBF, 1.5 * total_cpu), (total, total_cpu)
    for k, e in errs.items():
        assert e < max(TOLBF, 2.0 * errs_cpu[k]), (k, e, errs_cpu[k])


@pytest.mark.parametrize("mixed", [False, True])
def test_generator256_vs_oracle(dev, mixed):
    """StyleGan2Generator(256, base_dim 16) -- bench.py's G -- batch 4 with recorded noise: image and every
    parameter gradient against the oracle in fp64 (literal per-sample-weight grouped convolution).
    fp32: 1e-5, or 4x what the SAME oracle run in fp32 on the CPU achieves against its fp64 run on that tensor
    (weight gradients at 256x256 are 262 144-term sums).
    bf16: 1e-2 on the image and on the gradient as one vector (north_star's figure).  Per tensor the bound is DERIVED: the
    reference keeps ModulatedConv2d in fp32 under autocast and so offers no bf16 run of the generator, but the
    fp64 oracle can be told to STORE activations, their gradients and the convolution weights in bf16 while computing in fp64
    (oracle.gan_oracle.STORE_DTYPE) -- the rounding any bf16-storage implementation has, with none of its arithmetic error.
    A tensor passes at 3e-2 or at 3x what that run is off by on the same tensor (the Noise.scale gradients -- sums over pixels
    of g * noise with random signs, only ~sqrt(N) of their terms -- are where it matters)."""
    from gan_amd.modelling import build_generator
    from oracle import gan_oracle as O
    from oracle.detparams import det_uniform
    from oracle.tape import RandTape

    state = _g_state()
    z = det_uniform((4, 128), 321) * 3.0
    gen = build_generator("stylegan2", img_size=256, base_dim=16)
    gen.load_state_dict(state)
    gen = gen.to(dev)
    torch.manual_seed(20)  # the noise draws below come from the device generator
    with RandTape() as rec:  # the HIP path draws (bf16-exact values under autocast); the oracle replays them
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
            img = gen(z.to(dev))
    wgt = det_uniform((4, 3, 256, 256), 322) + 0.75  # positive: the generator step's loss -mean(D(img)) has no sign mix either
    loss = (img.float() * wgt.to(dev)).sum() / 1000.0 + img.float().square().mean()
    loss.backward()
    got = {k: p.grad for k, p in gen.named_parameters()}

    def oracle(dtype, store=None):
        sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in state.items()}
        O.STORE_DTYPE = store
        try:
            img_o = O.generator(sd, z.to(dtype), [t.to(dtype) for t in rec.tape])
            ((img_o * wgt.to(dtype)).sum() / 1000.0 + img_o.square().mean()).backward()
        finally:
            O.STORE_DTYPE = None
        return img_o.detach(), {k: v.grad.double() for k, v in sd.items()}

    img_o, ref = oracle(torch.float64)
    errs, total = _tensor_errs(got, ref)
    worst = max(errs.items(), key=lambda kv: kv[1])
    msg = f"G256 mixed={mixed}: img {rel_err(img, img_o):.2e}, all grads {total:.2e}, worst tensor {worst[0]} {worst[1]:.2e}"
    if not mixed:
        _, cpu32 = oracle(torch.float32)
        floor, total_floor = _tensor_errs(cpu32, ref)
        print(msg + f"; fp32 CPU oracle's own: all grads {total_floor:.2e}, worst {max(floor.values()):.2e}")
        assert rel_err(img, img_o) < TOL
        assert total < max(TOL, 4.0 * total_floor), (total, total_floor)
        for k, e in errs.items():
            assert e < max(TOL, 4.0 * floor[k]), (k, e, floor[k])
    else:
        _, stored = oracle(torch.float64, store=torch.bfloat16)
        yard, total_yard = _tensor_errs(stored, ref)
        wk = max(yard.items(), key=lambda kv: kv[1])
        print(msg + f"; fp64 oracle with bf16 storage: all grads {total_yard:.2e}, worst tensor {wk[0]} {wk[1]:.2e}")
        assert img.dtype == torch.bfloat16
        assert rel_err(img, img_o) < TOLBF
        assert total < TOLBF, total
        for k, e in errs.items():
            assert e < max(3e-2, 3.0 * yard[k]), (k, e, yard[k])


# ------------------------------------------------------------------------------------------------------
def _trainer(dev, size, batch, gen_base, disc_base, g_state, d_state, mixed):
    from gan_amd.train_celeba import TrainConfig, Trainer

    cfg = TrainConfig(
        model="stylegan2", img_size=size, gen_kwargs={"base_dim": gen_base}, disc_kwargs={"base_dim": disc_base},
        method="hinge", regularizer="r1", batch_size=batch, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
        lr=2e-4, device="cuda", mixed_precision=mixed, channels_last=mixed,
    )
    tr = Trainer(cfg, dev)
    tr.gen.load_state_dict(g_state)
    tr.disc.load_state_dict(d_state)
    return tr


@pytest.mark.parametrize("size,batch,gen_base", [(256, 4, 16), (32, 8, 64)])
def test_train_step_bf16_vs_oracle(dev, size, batch, gen_base):
    """One whole iteration (D step with the R1 double backward, G step) under the README flags (bf16 autocast +
    channels_last) against the ORACLE's iteration (fp32 on the CPU) on the same parameters, reals and random draws:
    both losses and both mean logits within bf16 tolerance."""
    from oracle import gan_oracle as O
    from oracle.detparams import det_uniform
    from oracle.tape import RandTape

    g_state = _g_state(size, gen_base, seed=7)
    d_state = _d_state(size, 16, seed=8)
    reals = det_uniform((batch, 3, size, size), 77) * 2.0
    g_sd = {k: v.clone().requires_grad_(True) for k, v in g_state.items()}
    d_sd = {k: v.clone().requires_grad_(True) for k, v in d_state.items()}
    opt_g = torch.optim.Adam(list(g_sd.values()), 2e-4, betas=(0.0, 0.9))
    opt_d = torch.optim.Adam(list(d_sd.values()), 2e-4, betas=(0.0, 0.9))
    with RandTape() as rec:
        ref = O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals.clone(), method="hinge", reg="r1")
    tr = _trainer(dev, size, batch, gen_base, 16, g_state, d_state, mixed=True)
    with RandTape(rec.tape):
        logs = tr.train_step(lambda: reals.clone().to(dev))
    got = {"loss_d": float(logs["loss/d"]), "loss_g": float(logs["loss/g"]), "d_real": float(logs["d/real"]), "d_fake": float(logs["d/fake"])}
    print(f"bf16 step {size}x{size}: HIP {got} oracle {ref}")
    for k in got:
        want = ref[k]
        # 3e-2: the losses are means of a few logits, each the end of a 14-convolution bf16 chain
        assert math.isfinite(got[k]) and abs(got[k] - want) <= 3e-2 * max(1.0, abs(want)), (k, got[k], want)


# ------------------------------------------------------------------------------------------------------
def test_ema_module_schedule_golden(dev, golden):
    """gan_amd.utils.EMA on the reference's own EMA trace (utils.py:75-99; nn.Linear(4, 3), warmup 100): untouched
    copy before step 100, exact copy at 100, lerp after -- through the flat arenas and gan_ema_lerp."""
    from torch import nn

    from gan_amd.utils import EMA

    g = golden("small_ops")["ema"]
    lin = nn.Linear(4, 3)
    lin.load_state_dict(g["init"])
    lin = lin.to(dev)
    ema = EMA(lin, beta=0.999, warmup=100)
    drift = g["drift"].to(dev)
    for step in range(1, 104):
        with torch.no_grad():
            lin.weight.add_(drift)
            lin.bias.add_(0.001 * step)
        ema.step()
        snap = g.get(str(step))
        if snap is not None:
            assert int(ema.counter) == snap["counter"]
            for k, v in ema.ema_model.state_dict().items():
                assert torch.allclose(v.cpu(), snap["ema"][k], rtol=1e-6, atol=1e-7), (step, k)
            for k, v in lin.state_dict().items():
                assert torch.allclose(v.cpu(), snap["model"][k], rtol=1e-5, atol=1e-6), (step, k)


def test_ema_forward_sees_updated_weights(dev):
    """EMA.step writes the averaged weights through the flat arena (no parameter version moves): a forward of the
    EMA generator after a step must not reuse the packed weight images cached by a forward before it."""
    import gan_amd.functional as GF
    from gan_amd.modelling import build_generator
    from gan_amd.utils import EMA
    from oracle.tape import RandTape

    torch.manual_seed(1)
    gen = build_generator("stylegan2", img_size=16, base_dim=128).to(dev)
    ema = EMA(gen, warmup=2)
    z = torch.randn(4, 128, device=dev)
    with RandTape() as rec, torch.no_grad():
        ema(z)  # packs and caches the EMA copy's weights
    for _ in range(3):  # copy at step 2, lerp at step 3
        with torch.no_grad():
            for p in gen.parameters():
                p.add_(0.05 * torch.randn_like(p))
        ema.step()
    with RandTape(rec.tape), torch.no_grad():
        after = ema(z)
    GF.drop_packed()
    with RandTape(rec.tape), torch.no_grad():
        fresh = ema(z)
    assert torch.equal(after, fresh)


# ------------------------------------------------------------------------------------------------------
def test_resample_convs_golden(dev, golden):
    """up_conv_blur / blur_conv_down (nvidia_ops.py:81-123): every branch of each, forward, first-order gradients
    of all leaves, second order through x, against the reference's own outputs."""
    from gan_amd.modelling import nvidia_ops

    for tag, g in golden("resample_convs").items():
        m = getattr(nvidia_ops, g["fn"])(6, 10, g["kernel_size"], g["blur_size"])
        assert list(m.state_dict().keys()) == list(g["state"].keys()), tag
        m.load_state_dict(g["state"])
        m = m.to(dev)
        x = g["x"].to(dev).requires_grad_(True)
        y = m(x)
        assert y.shape == g["y"].shape, tag
        assert rel_err(y, g["y"]) < TOL, (tag, rel_err(y, g["y"]))
        names = list(g["grad"].keys())
        params = dict(m.named_parameters())
        leaves = [x] + [params[n] for n in names[1:]]
        g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
        for n, t in zip(names, g1):
            assert rel_err(t, g["grad"][n]) < TOL, (tag, n, rel_err(t, g["grad"][n]))
        g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
        for n, t in zip(names, g2):
            ref = g["grad2"][n]
            if ref is None or float(ref.abs().max()) == 0.0:
                assert t is None or float(t.abs().max()) < 1e-6, (tag, n)
            else:
                assert rel_err(t, ref) < TOL2, (tag, n, rel_err(t, ref))
