"""GPU parity of the module API (ModulatedConv2d, generator, discriminator, R1 / WGAN-GP, DCGAN, the
train step) against the reference-generated golden vectors and against the oracle.

Tolerances (north_star): rel-err <= 1e-5 fp32 forward and first-order gradients, 1e-4 for second-order
quantities and multi-step trajectories, 1e-2 in bf16."""
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


def _load(module, sd, dev):
    module.load_state_dict(sd)
    return module.to(dev)


def _check_param_grads(module, ref_grads, tol):
    named = dict(module.named_parameters())
    scale = max(float(r.abs().max()) for r in ref_grads.values() if r is not None)
    for k, ref in ref_grads.items():
        got = named[k].grad
        if ref is None or float(ref.abs().max()) < 1e-7 * scale:
            assert got is None or float(got.abs().max()) < 1e-5 * scale, k
        else:
            assert got is not None, k
            assert rel_err(got, ref) < tol, (k, rel_err(got, ref))


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,k,demod", [("k3_demod", 3, True), ("k1_plain", 1, False)])
def test_modulated_conv_golden(dev, golden, tag, k, demod):
    from gan_amd.modelling.stylegan2 import ModulatedConv2d

    g = golden("modconv")[tag]
    m = _load(ModulatedConv2d(16, 32, k, 128, demodulation=demod), g["state"], dev)
    x = g["x"].to(dev).requires_grad_(True)
    w = g["w"].to(dev).requires_grad_(True)
    params = dict(m.named_parameters())
    names = ["x", "w", "weight", "bias", "style_weight.weight", "style_weight.bias"]
    leaves = [x, w] + [params[n] for n in names[2:]]
    y = m(x, w)
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < TOL, rel_err(y, g["y"])
    # first order, plain backward (hand-written non-differentiable path)
    g1 = torch.autograd.grad(y.square().sum(), leaves, retain_graph=True)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < TOL, (n, rel_err(t, g["grad"][n]))
    # first order under create_graph (differentiable path) and second order
    g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < TOL, (n, rel_err(t, g["grad"][n]))
    g2 = torch.autograd.grad(g1[0].square().sum() + g1[1].square().sum(), leaves, allow_unused=True)
    for n, t in zip(names, g2):
        ref = g["grad2"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert t is None or float(t.abs().max()) < 1e-6, n
        else:
            assert rel_err(t, ref) < TOL2, (n, rel_err(t, ref))


def test_batched_conv2d_golden(dev, golden):
    """`batched_conv2d` (stylegan2.py:18, vmap(F.conv2d)) fed the per-sample weights exactly as the reference's
    ModulatedConv2d.forward builds them (stylegan2.py:30-36) must reproduce the reference module's output, and its
    gradient with respect to x."""
    import torch.nn.functional as F

    from gan_amd.modelling.stylegan2 import batched_conv2d

    g = golden("modconv")["k3_demod"]
    st = {k: v.to(dev) for k, v in g["state"].items()}
    x = g["x"].to(dev).requires_grad_(True)
    w = g["w"].to(dev)
    style = F.linear(w, st["style_weight.weight"], st["style_weight.bias"]) + 1
    wmod = st["weight"].unsqueeze(0) * style.view(-1, 1, style.shape[1], 1, 1)
    wmod = F.normalize(wmod, dim=(2, 3, 4), eps=1e-8)
    y = batched_conv2d(x, wmod, st["bias"].expand(x.shape[0], -1), padding=1)
    assert rel_err(y, g["y"]) < TOL
    (gx,) = torch.autograd.grad(y.square().sum(), x)
    assert rel_err(gx, g["grad"]["x"]) < TOL


def test_modulated_conv_bf16_autocast(dev, golden):
    """Under bf16 autocast the kernel computes in bf16 with fp32 accumulation; tolerance 1e-2 against the
    fp32 reference output (the reference's ModulatedConv2d returns fp32 under autocast)."""
    from gan_amd.modelling.stylegan2 import ModulatedConv2d

    g = golden("modconv")["k3_demod"]
    m = _load(ModulatedConv2d(16, 32, 3, 128), g["state"], dev)
    x = g["x"].to(dev).requires_grad_(True)
    w = g["w"].to(dev).requires_grad_(True)
    params = dict(m.named_parameters())
    names = ["x", "w", "weight", "bias", "style_weight.weight", "style_weight.bias"]
    leaves = [x, w] + [params[n] for n in names[2:]]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x, w)
    assert y.dtype == torch.bfloat16
    assert rel_err(y, g["y"]) < TOLBF
    g1 = torch.autograd.grad(y.float().square().sum(), leaves)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < 3 * TOLBF, (n, rel_err(t, g["grad"][n]))


def test_noise_and_mbstd_modules(dev, golden):
    from gan_amd.modelling.progressive_gan import MinibatchStdDev
    from gan_amd.modelling.stylegan import Noise
    from oracle.tape import RandTape

    g = golden("small_ops")
    n = g["noise"]
    mod = Noise(n["scale"].shape[0]).to(dev)
    with torch.no_grad():
        mod.scale.copy_(n["scale"])
    with RandTape([n["draw"]]):
        assert rel_err(mod(n["x"].to(dev)), n["y"]) < TOL
    m = g["mbstd"]
    x = m["x"].to(dev).requires_grad_(True)
    y = MinibatchStdDev()(x)
    assert rel_err(y, m["y"]) < TOL
    (g1,) = torch.autograd.grad((y * m["wgt"].to(dev)).sum() + y[:, -1].square().sum(), x, create_graph=True)
    assert rel_err(g1, m["g1"]) < TOL
    (g2,) = torch.autograd.grad(g1.square().sum(), x)
    assert rel_err(g2, m["g2"]) < TOL2
    from gan_amd._lib import GanAmdError

    with pytest.raises(GanAmdError):
        MinibatchStdDev()(torch.zeros(6, 8, 4, 4, device=dev))  # batch not a multiple of the group size


def test_disc_block_golden(dev, golden):
    from gan_amd.modelling.progressive_gan import ProgressiveGanDiscriminatorBlock

    g = golden("disc")["block"]
    cin = g["state"]["layers.1.weight"].shape[0]
    cout = g["state"]["layers.3.weight"].shape[0]
    blk = _load(ProgressiveGanDiscriminatorBlock(cin, cout), g["state"], dev)
    x = g["x"].to(dev).requires_grad_(True)
    y = blk(x)
    assert rel_err(y, g["y"]) < TOL
    names = list(g["grad"].keys())
    params = dict(blk.named_parameters())
    leaves = [x] + [params[n] for n in names[1:]]
    g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < TOL, (n, rel_err(t, g["grad"][n]))
    g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
    for n, t in zip(names, g2):
        ref = g["grad2"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert t is None or float(t.abs().max()) < 1e-6, n
        else:
            assert rel_err(t, ref) < TOL2, (n, rel_err(t, ref))


def test_discriminator16_hinge_r1_gp_golden(dev, golden):
    from gan_amd.losses import get_loss, get_regularizer
    from gan_amd.modelling import build_discriminator
    from oracle.tape import RandTape

    g = golden("disc")["disc16"]
    base = g["state"]["0.weight"].shape[0]
    disc = _load(build_discriminator("stylegan2", img_size=16, base_dim=base), g["state"], dev)
    reals, fakes = g["reals"].to(dev), g["fakes"].to(dev)
    hinge = get_loss("hinge")
    assert rel_err(disc(reals), g["logits_reals"]) < TOL

    loss = hinge.d_loss(disc(reals), disc(fakes))
    loss.backward()
    assert math.isclose(float(loss), g["hinge"]["value"], rel_tol=1e-5)
    _check_param_grads(disc, g["hinge"]["grads"], TOL2)

    disc.zero_grad()
    r = reals.clone().requires_grad_()
    r1 = get_regularizer("r1")(disc, r, fakes, disc(r))
    r1.backward()
    assert math.isclose(float(r1), g["r1"]["value"], rel_tol=1e-5)
    _check_param_grads(disc, g["r1"]["grads"], TOL2)
    assert rel_err(r.grad, g["r1"]["grad_reals"]) < TOL2

    disc.zero_grad()
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    total = hinge.d_loss(d_r, disc(fakes)) + get_regularizer("r1")(disc, r, fakes, d_r)
    total.backward()
    assert math.isclose(float(total), g["hinge_r1"]["value"], rel_tol=1e-5)
    _check_param_grads(disc, g["hinge_r1"]["grads"], TOL2)

    disc.zero_grad()
    with RandTape([g["wgan_gp"]["alpha"]]):
        gp = get_regularizer("wgan-gp")(disc, reals, fakes, None)
    gp.backward()
    assert math.isclose(float(gp), g["wgan_gp"]["value"], rel_tol=1e-5)
    _check_param_grads(disc, g["wgan_gp"]["grads"], TOL2)


def test_generator16_golden(dev, golden):
    from gan_amd.modelling import build_generator
    from oracle import gan_oracle as O
    from oracle.detparams import det_state, det_uniform, projections
    from oracle.tape import RandTape

    g = golden("generator16")
    shapes = O.stylegan2_generator_shapes(g["img_size"], base_dim=g["base_dim"])
    gen = _load(build_generator("stylegan2", img_size=g["img_size"], base_dim=g["base_dim"]), det_state(shapes, seed=g["seed"]), dev)
    with RandTape(list(g["noise"])):
        img = gen(g["z"].to(dev))
    assert rel_err(img, g["img"]) < TOL, rel_err(img, g["img"])
    loss = (img * det_uniform(tuple(img.shape), 902).to(dev)).sum() + img.square().mean()
    assert math.isclose(float(loss), g["loss"], rel_tol=1e-5)
    loss.backward()
    named = dict(gen.named_parameters())
    for k, ref in g["grads"].items():
        got = named[k].grad
        if named[k].numel() > 4096:
            got = projections(got)
            scale = float(ref[1])
            assert float((got - ref).abs().max()) < 2e-5 * scale, (k, got, ref)
        else:
            assert rel_err(got, ref) < TOL2, (k, rel_err(got, ref))
    # the golden holds only 4-number fingerprints of the big tensors: compare every gradient in full
    # against the oracle (itself pinned to the same golden) on the same inputs
    # (oracle in fp64: its fp32 CPU run is itself 2.5e-3 off on learned_input's gradient, a 4608-deep sum)
    sd = {k: v.double().requires_grad_(True) for k, v in det_state(shapes, seed=g["seed"]).items()}
    img_o = O.generator(sd, g["z"].double(), [t.double() for t in g["noise"]])
    ((img_o * det_uniform(tuple(img_o.shape), 902).double()).sum() + img_o.square().mean()).backward()
    for k, v in sd.items():
        assert rel_err(named[k].grad, v.grad) < TOL, (k, rel_err(named[k].grad, v.grad))


def test_dcgan_wgan_gp_golden(dev, golden):
    from gan_amd.losses import get_loss, get_regularizer
    from gan_amd.modelling import build_discriminator, build_generator
    from oracle.tape import RandTape

    g = golden("dcgan")
    base_d = g["d_state"]["0.0.weight"].shape[0]
    disc = _load(build_discriminator("dcgan", img_size=64, base_dim=base_d, norm="none"), g["d_state"], dev)
    base_g = g["g_state"]["4.0.weight"].shape[0] if "4.0.weight" in g["g_state"] else None
    gen = build_generator("dcgan", img_size=64, base_dim=base_g or 8)
    gen = _load(gen, g["g_state"], dev)
    wgan = get_loss("wgan")
    fakes = gen(g["z"].to(dev))
    assert rel_err(fakes, g["fakes"]) < TOL
    reals = g["reals"].to(dev)
    assert rel_err(disc(reals), g["logits_reals"]) < TOL
    with RandTape([g["wgan_gp"]["alpha"]]):
        gp = get_regularizer("wgan-gp")(disc, reals, fakes.detach(), None)
    loss = wgan.d_loss(disc(reals), disc(fakes.detach())) + gp
    loss.backward()
    assert math.isclose(float(gp), g["wgan_gp"]["value"], rel_tol=1e-5)
    assert math.isclose(float(loss), g["wgan_gp"]["loss"], rel_tol=1e-5)
    _check_param_grads(disc, g["wgan_gp"]["grads"], TOL2)
    disc.zero_grad()
    gl = wgan.g_loss(disc(gen(g["z"].to(dev))), None)
    gl.backward()
    assert math.isclose(float(gl), g["g_loss"]["value"], rel_tol=1e-5)
    _check_param_grads(gen, g["g_loss"]["grads"], TOL2)


# ------------------------------------------------------------------------------------------
def _trainer16(dev, g_state, d_state, mixed=False, gen_base=128, disc_base=16, size=16, batch=4, warmup=100):
    from gan_amd.train_celeba import TrainConfig, Trainer

    cfg = TrainConfig(
        model="stylegan2", img_size=size, gen_kwargs={"base_dim": gen_base}, disc_kwargs={"base_dim": disc_base},
        method="hinge", regularizer="r1", batch_size=batch, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
        lr=2e-4, device="cuda", mixed_precision=mixed, channels_last=mixed,
    )
    tr = Trainer(cfg, dev)
    # load_state_dict copies in place: the flat arenas stay intact
    tr.gen.load_state_dict(g_state)
    tr.disc.load_state_dict(d_state)
    tr.gen_ema.ema_model.load_state_dict(g_state)
    tr.gen_ema.warmup = warmup
    return tr


def test_trajectory16_golden(dev, golden):
    """10 free-running hinge + R1 iterations (Adam 2e-4, betas (0, 0.9)) against the losses the reference
    itself produced, with the reference's recorded random draws.  The iteration is chaotic (see
    oracle/make_envelope.py: the oracle in fp32 and in fp64 drifts apart by O(1) within 10 steps), so
    each step is held to 1e-4 relative OR 10x the recorded fp32-vs-fp64 envelope of that step, measured
    from the nearer of the two recorded trajectories (observed: <= 3.3x; the fp32 atomics of the
    small-shape reductions make the GPU run itself vary from launch to launch, and the iteration
    amplifies that too).  Inside the predictability horizon the bounds are strict: steps 0-2 against
    the reference's fp32 losses, steps 0-3 against the fp64 oracle's, which the HIP path follows for
    longer than the reference's own fp32 run does (step 3: 2e-5 against 1.4e-3)."""
    from oracle import gan_oracle as O
    from oracle.detparams import det_state, det_uniform, projections
    from oracle.tape import RandTape

    g = golden("trajectory16")
    env = golden("trajectory16_envelope")["envelope"]
    traj64 = golden("trajectory16_envelope")["fp64"]
    g_state = det_state(O.stylegan2_generator_shapes(16, base_dim=128), seed=g["gen_seed"], bias_amp=0.0, scale_amp=0.05)
    d_state = det_state(O.stylegan2_discriminator_shapes(16, base_dim=16), seed=g["disc_seed"], bias_amp=0.0, scale_amp=0.5)
    tr = _trainer16(dev, g_state, d_state, warmup=g["ema_warmup"])
    for step in range(10):
        reals = (det_uniform((4, 3, 16, 16), g["reals_salt0"] + step) * 2.0).to(dev)
        with RandTape(g["tape"][step]):
            logs = tr.train_step(lambda: reals.clone())
        ref = g["steps"][step]
        for j, key in enumerate(("loss/d", "loss/g")):
            got, want = float(logs[key]), float(ref[j])
            f64 = float(traj64[step, j])
            tol = max(1e-4 * max(1.0, abs(want)), 10.0 * float(env[step, j]))
            print(f"trajectory16 step {step} {key}: |got - reference fp32| {abs(got - want):.2e}, |got - oracle fp64| {abs(got - f64):.2e}, envelope {float(env[step, j]):.2e}")
            assert min(abs(got - want), abs(got - f64)) <= tol, (step, key, got, want, f64, tol)
            if step <= 3:
                assert math.isclose(got, f64, rel_tol=2e-4, abs_tol=2e-4), (step, key, got, f64)
        if step < 3:  # inside the predictability horizon: strict
            assert math.isclose(float(logs["loss/d"]), float(ref[0]), rel_tol=1e-4, abs_tol=1e-5), (step, logs, ref)
            assert math.isclose(float(logs["loss/g"]), float(ref[1]), rel_tol=1e-4, abs_tol=1e-5), (step, logs, ref)
    # after 10 chaotic steps only coarse agreement is meaningful: every weight moved by <= 10 Adam steps
    # (weight matrices only: biases start at zero and are pure Adam random walks by then)
    named = dict(tr.gen.named_parameters())
    for k, ref in g["final"]["gen"].items():
        if named[k].ndim >= 2 and k.endswith("weight"):
            got = projections(named[k])
            assert abs(float(got[1] - ref[1])) < 2e-2 * max(float(ref[1]), 1e-3), k
    ema_named = dict(tr.gen_ema.ema_model.named_parameters())
    for k, ref in g["final"]["ema"].items():
        if ema_named[k].ndim >= 2 and k.endswith("weight"):
            got = projections(ema_named[k])
            assert abs(float(got[1] - ref[1])) < 2e-2 * max(float(ref[1]), 1e-3), k


def _recipe_trainer(dev, g, **over):
    from gan_amd.train_celeba import TrainConfig, Trainer
    from oracle.detparams import recipe_state

    kw = dict(model=g["model"], img_size=g["img_size"], gen_kwargs=dict(g["gen_kwargs"]), disc_kwargs=dict(g["disc_kwargs"]), method=g["method"],
              regularizer=g["regularizer"], n_disc=g["n_disc"], batch_size=g["batch"], optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
              lr=g["lr"], device="cuda", mixed_precision=False, channels_last=False)
    kw.update(over)
    tr = Trainer(TrainConfig(**kw), dev)
    g_state = recipe_state(g["gen_shapes"], g["gen_seed"], 0.05, g["wgain"])
    tr.gen.load_state_dict(g_state)
    tr.disc.load_state_dict(recipe_state(g["disc_shapes"], g["disc_seed"], 0.5, g["wgain"]))
    tr.gen_ema.ema_model.load_state_dict(g_state)
    tr.gen_ema.warmup = g["ema_warmup"]
    return tr


@pytest.mark.parametrize("name", ["dcgan_wgan_gp", "dcgan_wgan_clip", "stylegan2_64"])
def test_trainer_recipes_golden(dev, golden, name):
    """The Trainer on the README recipes the 16 x 16 trajectory does not reach, against the REFERENCE's own modules run
    through the train_celeba.py:111-149 loop (oracle/make_goldens.py: gold_trainer_recipes): DCGAN WGAN-GP with n_disc = 5
    (five gradient-penalty double backwards per iteration), DCGAN WGAN with weight clipping (:115-118), StyleGAN2 at
    64 x 64 / base_dim 64 with hinge + R1 (README.md:55,60).  Every D loss and the G loss of every iteration with the
    reference's recorded draws, then a fingerprint of every parameter (fp32, launched from the host)."""
    from oracle.detparams import det_uniform, projections
    from oracle.tape import RandTape

    g = golden("trainer_recipes")[name]
    tr = _recipe_trainer(dev, g)
    salt = [g["reals_salt0"]]

    def next_reals():
        salt[0] += 1
        return (det_uniform((g["batch"], 3, g["img_size"], g["img_size"]), salt[0]) * 2.0).to(dev)

    for step, ref in enumerate(g["steps"]):
        with RandTape(g["tape"][step]):
            logs = tr.train_step(next_reals)
        got = [float(v) for v in logs["loss/d_steps"]] + [float(logs["loss/g"])]
        assert len(got) == len(ref)
        # a WGAN loss is a difference of terms of size `scale`, and the trajectory amplifies a rounding difference about
        # 4x per iteration (n_disc Adam steps with beta1 = 0 are sign-like updates): the CPU oracle, fp32 like the
        # reference, sits 9e-5 from it at the third iteration, the MFMA path (other summation order) 2.6e-3
        grow = 4.0 ** step
        scale = max(1.0, float(ref.abs().max())) * grow
        for j, (a, b) in enumerate(zip(got, ref.tolist())):
            assert math.isclose(a, b, rel_tol=2e-4 * grow, abs_tol=2e-5 * scale), (name, step, j, a, b)
    for mod, key in ((tr.gen, "gen"), (tr.disc, "disc"), (tr.gen_ema.ema_model, "ema")):
        sd = mod.state_dict()
        for k, ref in g["final"][key].items():
            if not torch.is_floating_point(sd[k]):
                continue
            got = projections(sd[k])
            assert float((got - ref).abs().max()) < 5e-4 * 4.0 ** (len(g["steps"]) - 1) * max(float(ref[1]), 1e-3), (name, key, k, got, ref)


def test_trainer_graph_matches_host_launches_n_disc_and_grad_accum(dev, golden):
    """The iteration replayed as HIP graphs against the same iteration launched from the host, where a step has several
    discriminator passes: n_disc = 2 (every D step's gradient cut is followed by the generator forward the next pass
    needs), grad_accum = 2 (the gradient arena keeps accumulating across two backward passes before the exchange), and the
    deferred generator update of multi-rank runs (its all-reduce left in flight over the iteration boundary; the EMA copy
    is compared too).  DCGAN WGAN-GP; same seed, so the
    device generator hands both trainers the same draws; five iterations (two of them before the capture), four with Adam.
    SGD: the update is proportional to the gradient, so everything agrees to rounding.  Adam with beta1 = 0 turns the
    rounding-sized gradient components behind the generator's BatchNorm (the direction that rescales a channel has zero
    gradient) into lr-sized steps: there both networks' parameters are held to the size of a step and the losses to 5e-3 instead (two host-launched runs differ by as much: the
    split-K reductions add in arrival order, and the fifth iteration is 16x further apart than the fourth)."""
    g = golden("trainer_recipes")["dcgan_wgan_gp"]
    for over in (dict(n_disc=2), dict(n_disc=1, grad_accum=2), dict(n_disc=2, optimizer="SGD", optimizer_kwargs={}, lr=1e-3),
                 dict(n_disc=1, grad_accum=2, optimizer="SGD", optimizer_kwargs={}, lr=1e-3),
                 dict(n_disc=2, defer_g_step=True, optimizer="SGD", optimizer_kwargs={}, lr=1e-3), dict(n_disc=1, defer_g_step=True)):
        runs = []
        for graph in (False, True):
            torch.manual_seed(77)
            tr = _recipe_trainer(dev, g, hip_graph=graph, graph_warmup=2, **over)
            micro = g["batch"] // over.get("grad_accum", 1)
            reals = [torch.randn(micro, 3, g["img_size"], g["img_size"], device=dev) for _ in range(4)]
            it = [0]

            def next_reals():
                it[0] += 1
                return reals[it[0] % 4].clone()

            losses = []
            for _ in range(5 if over.get("optimizer") == "SGD" else 4):
                logs = tr.train_step(next_reals)
                losses.append([float(v) for v in logs["loss/d_steps"]] + [float(logs["loss/g"])])
            if graph:
                assert tr._graph is not None
                cuts = [cut[0] for _, cut in tr._graph if cut is not None]
                want = ["start", "wait"] * over["n_disc"] + ["reduce"]
                if over.get("defer_g_step"):
                    want = ["finish_g"] + want[:-1] + ["start_g"]
                assert cuts == want, cuts
                tr.flush()  # (the deferred generator update of the last iteration)
            gsd = {k: v.detach().clone() for k, v in tr.gen.state_dict().items()}
            gsd.update({"ema." + k: v.detach().clone() for k, v in tr.gen_ema.ema_model.state_dict().items()})
            runs.append((losses, {k: v.detach().clone() for k, v in tr.disc.state_dict().items()}, gsd))
            assert int(tr.gen_ema.counter) == len(losses)
        (l0, d0, g0), (l1, d1, g1) = runs
        for a, b in zip(l0, l1):
            assert len(a) == over["n_disc"] * over.get("grad_accum", 1) + 1
            tol = 1e-4 if over.get("optimizer") == "SGD" else 5e-3
            for x, y in zip(a, b):
                assert math.isclose(x, y, rel_tol=tol, abs_tol=tol), (over, l0, l1)
        for s0, s1 in ((d0, d1), (g0, g1)):
            for k in s0:
                if not torch.is_floating_point(s0[k]):
                    continue
                if over.get("optimizer") == "SGD":
                    assert rel_err(s1[k], s0[k]) < 1e-4, (over, k)
                else:
                    diff = (s1[k] - s0[k]).abs()  # (a flipped sign moves an element by 2 lr per optimizer step; few flip)
                    assert float(diff.max()) < 16 * g["lr"] and float(diff.mean()) < g["lr"], (over, k)


def test_trainer_grad_accum_accumulates(dev, golden):
    """grad_accum = 2 (train_celeba.py:120-135): the two backward passes of a discriminator step ADD their gradients (each
    loss is the mean over its own micro-batch -- the reference does not rescale) before one optimizer step.  With SGD the
    parameter change is -lr x that sum: compared with the two gradients computed one by one through the same modules."""
    from gan_amd.losses import get_loss, get_regularizer
    from oracle.tape import RandTape

    g = golden("trainer_recipes")["dcgan_wgan_gp"]
    torch.manual_seed(78)
    tr = _recipe_trainer(dev, g, grad_accum=2, optimizer="SGD", optimizer_kwargs={}, lr=0.5, n_disc=1)  # (a step well above fp32 spacing)
    micro = g["batch"] // 2
    batches = [torch.randn(micro, 3, 64, 64, device=dev) for _ in range(2)]
    before = {k: v.detach().clone() for k, v in tr.disc.named_parameters()}
    # the expected gradient sum, pass by pass, with the draws recorded for the trainer to replay
    crit, reg = get_loss("wgan"), get_regularizer("wgan-gp")
    expect = {k: torch.zeros_like(v) for k, v in before.items()}
    with RandTape() as rec:
        for reals in batches:
            reals = reals.clone().requires_grad_()
            zs = torch.randn(micro, 128, device=dev)
            with torch.no_grad():
                fakes = tr.gen(zs)
            d_reals, d_fakes = tr.disc(reals), tr.disc(fakes)
            loss = crit.d_loss(d_reals, d_fakes) + reg(tr.disc, reals, fakes, d_reals)
            grads = torch.autograd.grad(loss, list(tr.disc.parameters()))
            for (k, _), gr in zip(tr.disc.named_parameters(), grads):
                expect[k] += gr
        for _ in range(2):  # the generator step's draws
            torch.randn(micro, 128, device=dev)
    feed = iter(batches)
    with RandTape(rec.tape):
        tr.train_step(lambda: next(feed).clone())
    for k, p in tr.disc.named_parameters():
        upd = p.detach() - before[k]
        assert rel_err(upd, -0.5 * expect[k]) < 1e-4, (k, rel_err(upd, -0.5 * expect[k]))


def _sync(module, sd):
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(sd[k].detach())


def test_loss_match_100_steps_vs_oracle(dev):
    """north_star: matching G/D losses over 100 steps.  The oracle runs 100 hinge + R1 iterations on the CPU
    (fp32, reference init, README optimiser).  Because the free-running iteration is chaotic in fp32 (see
    test_trajectory16_golden), the HIP trainer is re-synchronised to the oracle's parameters before every
    iteration (its own Adam moments and EMA keep running): every one of the 100 iterations -- D step with
    R1 double backward, Adam, G step, Adam, EMA -- is then compared on the oracle's trajectory: both
    losses to 1e-4 and the parameter update of both networks in relative L2."""
    from oracle import gan_oracle as O
    from oracle.detparams import det_uniform
    from oracle.tape import RandTape

    size, batch, steps = 16, 4, 100
    torch.manual_seed(3)
    g_sd, d_sd = O.new_stylegan2_state(size, 128, 16)
    with torch.no_grad():  # exercise the noise inputs too (reference init has Noise.scale = 0)
        for k, v in g_sd.items():
            if k.endswith(".scale"):
                v.add_(0.05 * torch.randn_like(v))
    opt_d = torch.optim.Adam(list(d_sd.values()), 2e-4, betas=(0.0, 0.9))
    opt_g = torch.optim.Adam(list(g_sd.values()), 2e-4, betas=(0.0, 0.9))
    tr = _trainer16(dev, {k: v.detach() for k, v in g_sd.items()}, {k: v.detach() for k, v in d_sd.items()}, warmup=10)
    ema = [v.detach().clone() for v in g_sd.values()]
    counter = 0
    worst = dict(d=0.0, g=0.0, ud=0.0, ug=0.0)
    for step in range(steps):
        before = {id(sd): {k: v.detach().clone() for k, v in sd.items()} for sd in (g_sd, d_sd)}
        _sync(tr.gen, g_sd)
        _sync(tr.disc, d_sd)
        reals = det_uniform((batch, 3, size, size), 4000 + step) * 2.0
        with RandTape() as rec:
            ref = O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals.clone(), method="hinge", reg="r1")
        counter = O.ema_step(ema, list(g_sd.values()), counter, warmup=10)
        with RandTape(rec.tape):
            logs = tr.train_step(lambda: reals.clone().to(dev))
        worst["d"] = max(worst["d"], abs(float(logs["loss/d"]) - ref["loss_d"]) / max(1.0, abs(ref["loss_d"])))
        worst["g"] = max(worst["g"], abs(float(logs["loss/g"]) - ref["loss_g"]) / max(1.0, abs(ref["loss_g"])))
        if step >= 1:  # Adam moments of the two sides agree from the second iteration on
            for key, sd, mod in (("ug", g_sd, tr.gen), ("ud", d_sd, tr.disc)):
                num = den = 0.0
                for k, v in mod.state_dict().items():
                    upd_ref = sd[k].detach() - before[id(sd)][k]
                    upd = v.detach().cpu() - before[id(sd)][k]
                    num += float((upd - upd_ref).double().square().sum())
                    den += float(upd_ref.double().square().sum())
                worst[key] = max(worst[key], (num / max(den, 1e-30)) ** 0.5)
    print(f"100-step match (re-synchronised): worst rel loss diff D {worst['d']:.2e} G {worst['g']:.2e}; "
          f"worst rel L2 update diff D {worst['ud']:.2e} G {worst['ug']:.2e}")
    assert worst["d"] < 1e-4 and worst["g"] < 1e-4, worst
    assert worst["ud"] < 5e-2 and worst["ug"] < 5e-2, worst
    # EMA (utils.py:84-96; warmup 10: copy at the 10th step, 90 lerps after): the HIP trainer averaged its own
    # post-step generator weights, the oracle its own -- they differ by the update differences bounded above
    # (<= 5e-2 of a 2e-4 Adam step), so the two averages agree to ~1e-5 absolute
    assert counter == steps and int(tr.gen_ema.counter) == steps
    for (k, v), ref in zip(tr.gen_ema.ema_model.state_dict().items(), ema):
        assert float((v.detach().cpu() - ref).abs().max()) < 5e-5, k


def test_bucket_allreduce_on_rccl_single_rank(dev, monkeypatch):
    """The data-parallel gradient path on the real backend: a 1-rank RCCL group with the bucket all-reduces
    forced on (GAN_AMD_FORCE_ALLREDUCE) must reproduce the plain single-process iteration exactly -- the
    async all-reduces of arena slices fired from the post-accumulate hooks, and their hand-over to the
    optimizer's stream, are the same code the N-GPU runs use (multi-rank arithmetic: gloo test on CPU)."""
    import torch.distributed as dist
    from gan_amd.train_celeba import TrainConfig, Trainer

    def run(force):
        if force:
            monkeypatch.setenv("GAN_AMD_FORCE_ALLREDUCE", "1")
        else:
            monkeypatch.delenv("GAN_AMD_FORCE_ALLREDUCE", raising=False)
        torch.manual_seed(7)
        cfg = TrainConfig(model="stylegan2", img_size=32, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16},
                          method="hinge", regularizer="r1", batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
                          lr=2e-4, device="cuda", mixed_precision=False, channels_last=False)
        tr = Trainer(cfg, dev)
        reals = torch.randn(8, 3, 32, 32, device=dev)
        out = []
        for _ in range(3):
            logs = tr.train_step(lambda: reals.clone())
            out.append((float(logs["loss/d"]), float(logs["loss/g"])))
        return out, [r._fired for r in (tr.red_d, tr.red_g)]

    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29731")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        forced, fired = run(True)
        plain, _ = run(False)
    finally:
        dist.destroy_process_group()
    assert all(all(f) for f in fired)
    for a, b in zip(forced, plain):
        assert abs(a[0] - b[0]) <= 1e-3 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-3 * max(1.0, abs(b[1])), (forced, plain)


def test_hip_graph_iteration_matches_eager(dev, monkeypatch):
    """cfg.hip_graph: the iteration replayed as HIP graphs (cut at the gradient all-reduces, which run between
    the replays on a 1-rank RCCL group here) against the same iteration launched from the host.  torch.randn
    is pinned to fixed draws so both see the same latents and noise."""
    import torch.distributed as dist
    from gan_amd.train_celeba import TrainConfig, Trainer

    real_randn = torch.randn
    cache = {}

    def fixed_randn(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        key = (shape, str(kw.get("dtype")), str(kw.get("device")))
        if key not in cache:
            g = torch.Generator(device="cpu").manual_seed(len(cache) + 11)
            cache[key] = real_randn(*shape, generator=g).to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)
        return cache[key]

    def run(graph):
        torch.manual_seed(7)
        cfg = TrainConfig(model="stylegan2", img_size=32, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16},
                          method="hinge", regularizer="r1", batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
                          lr=2e-4, device="cuda", mixed_precision=False, channels_last=False, hip_graph=graph, graph_warmup=1)
        tr = Trainer(cfg, dev)
        g = torch.Generator(device="cpu").manual_seed(3)
        pool = [real_randn(8, 3, 32, 32, generator=g).to(dev) for _ in range(5)]
        out = []
        monkeypatch.setattr(torch, "randn", fixed_randn)
        try:
            for i in range(4):  # the iteration is chaotic: a 1e-7 reduction-order difference is ~2e-2 by the 5th step
                logs = tr.train_step(lambda: pool[i])
                out.append((float(logs["loss/d"]), float(logs["loss/g"])))
        finally:
            monkeypatch.setattr(torch, "randn", real_randn)
        return out, tr

    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29733")
    monkeypatch.setenv("GAN_AMD_FORCE_ALLREDUCE", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        eager, _ = run(False)
        graphed, tr = run(True)
    finally:
        dist.destroy_process_group()
    assert tr._graph is not None and len(tr._graph) == 4  # D pass | G forward (under D's all-reduce) | D update + G pass | G update
    for i, (a, b) in enumerate(zip(graphed, eager)):
        assert abs(a[0] - b[0]) <= 1e-2 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-2 * max(1.0, abs(b[1])), (i, graphed, eager)


def test_hip_graph_capture_failure_falls_back_to_eager(dev, monkeypatch, capsys):
    """A capture that raises after its first segment was recorded (never executed): the trainer must drop what the
    aborted capture left behind -- packed weights and padded inputs in graph-pool buffers, handed-over gradient
    tensors -- and run that step and the following ones from the host, with the losses of a trainer that never tried."""
    from gan_amd.train_celeba import TrainConfig, Trainer

    real_randn = torch.randn
    cache = {}

    def fixed_randn(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        key = (shape, str(kw.get("dtype")), str(kw.get("device")))
        if key not in cache:
            g = torch.Generator(device="cpu").manual_seed(len(cache) + 17)
            cache[key] = real_randn(*shape, generator=g).to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)
        return cache[key]

    real_graph = torch.cuda.graph
    entered = {"n": 0}

    class FailingGraph(real_graph):
        def __enter__(self):
            entered["n"] += 1
            if entered["n"] == 2:
                raise RuntimeError("capture refused (test)")
            return super().__enter__()

    def run(graph):
        torch.manual_seed(9)
        cfg = TrainConfig(model="stylegan2", img_size=32, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16},
                          method="hinge", regularizer="r1", batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]},
                          lr=2e-4, device="cuda", mixed_precision=False, channels_last=False, hip_graph=graph, graph_warmup=1)
        tr = Trainer(cfg, dev)
        g = torch.Generator(device="cpu").manual_seed(5)
        pool = [real_randn(8, 3, 32, 32, generator=g).to(dev) for _ in range(4)]
        out = []
        monkeypatch.setattr(torch, "randn", fixed_randn)
        try:
            for i in range(4):
                logs = tr.train_step(lambda: pool[i])
                out.append((float(logs["loss/d"]), float(logs["loss/g"])))
        finally:
            monkeypatch.setattr(torch, "randn", real_randn)
        return out, tr

    eager, _ = run(False)
    monkeypatch.setattr(torch.cuda, "graph", FailingGraph)
    try:
        fallen, tr = run(True)
    finally:
        monkeypatch.setattr(torch.cuda, "graph", real_graph)
    assert entered["n"] == 2 and tr._graph is None and tr.cfg.hip_graph is False
    assert "continuing without graphs" in capsys.readouterr().err
    for i, (a, b) in enumerate(zip(fallen, eager)):
        assert abs(a[0] - b[0]) <= 1e-2 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-2 * max(1.0, abs(b[1])), (i, fallen, eager)


def test_trainer_with_diff_augment(dev):
    """TrainConfig.diff_augment (train_celeba.py:45,63): the discriminator wrapped in nn.Sequential(DiffAugment(), disc),
    hinge + R1 differentiating through the augmentation twice, launched from the host and replayed as HIP graphs."""
    from gan_amd.diff_augment import DiffAugment
    from gan_amd.train_celeba import TrainConfig, Trainer

    for graph in (False, True):
        torch.manual_seed(13)
        cfg = TrainConfig(model="stylegan2", img_size=32, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16}, method="hinge",
                          regularizer="r1", batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]}, lr=2e-4, device="cuda",
                          mixed_precision=True, channels_last=True, hip_graph=graph, graph_warmup=1, diff_augment=True)
        tr = Trainer(cfg, dev)
        assert isinstance(tr.disc[0], DiffAugment) and next(iter(tr.disc.state_dict())).startswith("1.")
        reals = torch.randn(8, 3, 32, 32, device=dev)
        before = [p.detach().clone() for p in tr.disc.parameters()]
        for _ in range(3):
            logs = tr.train_step(lambda: reals)
        assert all(math.isfinite(float(v)) for v in logs.values())
        assert any(not torch.equal(a, b) for a, b in zip(before, tr.disc.parameters()))
        assert (tr._graph is not None) == graph


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["stylegan", "progressive_gan"])
def test_generators_v1_golden(dev, golden, name):
    """build_generator("stylegan" | "progressive_gan") -- AdaIN / LayerNorm2d kernels, the unmodulated fused
    [Upsample ->] Conv -> Noise -> LeakyReLU launch, the RGB / block skips -- against the reference's image, loss and
    parameter gradients (fp32), every gradient in full against the fp64 oracle, and under bf16 autocast."""
    from gan_amd.modelling import build_generator
    from oracle import gan_oracle as O
    from oracle.detparams import det_state, det_uniform, projections
    from oracle.tape import RandTape

    g = golden("generators_v1")[name]
    shapes = {k: tuple(v) for k, v in g["shapes"].items()}
    state = det_state(shapes, seed=g["seed"], bias_amp=0.1, scale_amp=0.1)
    gen = build_generator(name, img_size=16, base_dim=16)
    assert {k: tuple(v.shape) for k, v in gen.state_dict().items()} == shapes
    gen = _load(gen, state, dev)
    wgt = det_uniform(tuple(g["img"].shape), 950 + g["seed"])
    with RandTape(list(g["noise"])):
        img = gen(g["z"].to(dev))
    assert rel_err(img, g["img"]) < TOL, rel_err(img, g["img"])
    loss = (img * wgt.to(dev)).sum() + img.square().mean()
    assert math.isclose(float(loss), g["loss"], rel_tol=1e-5)
    loss.backward()
    named = dict(gen.named_parameters())
    for k, ref in g["grads"].items():
        got = named[k].grad
        if named[k].numel() > 4096:
            assert float((projections(got) - ref).abs().max()) < 2e-4 * float(ref[1]), (k, projections(got), ref)
        else:
            assert rel_err(got, ref) < TOL2, (k, rel_err(got, ref))
    sd = {k: v.double().requires_grad_(True) for k, v in state.items()}
    noise64 = [t.double() for t in g["noise"]]
    img_o = O.stylegan_generator(sd, g["z"].double(), noise64) if name == "stylegan" else O.progan_generator(sd, g["z"].double())
    ((img_o * wgt.double()).sum() + img_o.square().mean()).backward()
    for k, v in sd.items():
        assert rel_err(named[k].grad, v.grad) < TOL2, (k, rel_err(named[k].grad, v.grad))
    # bf16 autocast: image within 1e-2, the gradient as one vector within 3e-2 (the oracle re-run on the noise draws as
    # the bf16 path sees them: the Noise layer draws in the activation dtype, stylegan.py:26)
    gen.zero_grad()
    with RandTape(list(g["noise"])), torch.autocast("cuda", dtype=torch.bfloat16):
        img_b = gen(g["z"].to(dev))
    sd = {k: v.double().requires_grad_(True) for k, v in state.items()}
    noise_b = [t.bfloat16().double() for t in g["noise"]]
    img_o = O.stylegan_generator(sd, g["z"].double(), noise_b) if name == "stylegan" else O.progan_generator(sd, g["z"].double())
    ((img_o * wgt.double()).sum() + img_o.square().mean()).backward()
    # yardstick: the same oracle under torch.autocast(cpu, bf16) -- the reference arithmetic's own bf16 error (instance /
    # layer normalisation subtract a mean: the rounding of their bf16 inputs is amplified by |mean| / std)
    sc = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        img_c = (O.stylegan_generator(sc, g["z"], [t.bfloat16().float() for t in g["noise"]]) if name == "stylegan" else O.progan_generator(sc, g["z"])).float()
    ((img_c * wgt).sum() + img_c.square().mean()).backward()

    def total(grads):
        num = sum(float((grads[k].double().cpu() - v.grad).square().sum()) for k, v in sd.items())
        return (num / sum(float(v.grad.square().sum()) for v in sd.values())) ** 0.5

    e_img, e_img_cpu = rel_err(img_b, img_o), rel_err(img_c, img_o)
    # 3x for the image: under autocast the reference's Noise output is fp32 (bf16 conv output + fp32 noise * scale) and
    # feeds AdaIN unrounded, here every activation is stored in bf16 -- one more rounding in front of a normalisation
    assert img_b.dtype == torch.bfloat16 and e_img < max(TOLBF, 3.0 * e_img_cpu), (e_img, e_img_cpu)
    ((img_b.float() * wgt.to(dev)).sum() + img_b.float().square().mean()).backward()
    e_g, e_g_cpu = total({k: p.grad for k, p in named.items()}), total({k: v.grad for k, v in sc.items()})
    print(f"{name} bf16: image {e_img:.2e} (CPU autocast {e_img_cpu:.2e}), all grads {e_g:.2e} (CPU autocast {e_g_cpu:.2e})")
    assert e_g < max(3e-2, 2.0 * e_g_cpu), (e_g, e_g_cpu)


# ------------------------------------------------------------------------------------------
def test_blur_enabled_discriminator(dev, golden):
    """SURVEY 8 f1, discriminator side.  (a) blur_size=2: Blur(2, down=2) IS AvgPool2d(2), so the FIR path of the block
    must reproduce the REFERENCE's discriminator goldens (logits, hinge + R1 value, every parameter gradient through the
    double backward of upfirdn2d).  (b) blur_size=4: against the oracle's composition in fp64 (no reference model has the
    blur: parity there is pinned by the pinned ops, not by a reference output)."""
    from gan_amd.losses import get_loss, get_regularizer
    from gan_amd.modelling import build_discriminator
    from oracle import gan_oracle as O

    g = golden("disc")["disc16"]
    base = g["state"]["0.weight"].shape[0]
    reals, fakes = g["reals"].to(dev), g["fakes"].to(dev)
    hinge, r1 = get_loss("hinge"), get_regularizer("r1")
    # (a)
    disc = build_discriminator("stylegan2", img_size=16, base_dim=base, blur_size=2)
    disc.load_state_dict(g["state"], strict=False)  # the Blur buffers are the only extra keys
    disc = disc.to(dev)
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    assert rel_err(d_r, g["logits_reals"]) < TOL
    total = hinge.d_loss(d_r, disc(fakes)) + r1(disc, r, fakes, d_r)
    total.backward()
    assert math.isclose(float(total), g["hinge_r1"]["value"], rel_tol=1e-5)
    _check_param_grads(disc, g["hinge_r1"]["grads"], TOL2)
    # (b)
    disc = build_discriminator("stylegan2", img_size=16, base_dim=base, blur_size=4)
    disc.load_state_dict(g["state"], strict=False)
    disc = disc.to(dev)
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    total = hinge.d_loss(d_r, disc(fakes)) + r1(disc, r, fakes, d_r)
    total.backward()
    sd = {k: v.double().requires_grad_(True) for k, v in g["state"].items()}
    ro = g["reals"].double().requires_grad_(True)
    o_r = O.discriminator(sd, ro, blur_size=4)
    want = O.d_loss("hinge", o_r, O.discriminator(sd, g["fakes"].double(), blur_size=4)) + O.r1_penalty(None, ro, None, o_r)
    want.backward()
    assert rel_err(d_r, o_r) < TOL and math.isclose(float(total), float(want), rel_tol=1e-5)
    _check_param_grads(disc, {k: v.grad.float() for k, v in sd.items()}, TOL2)


def test_blur_enabled_generator(dev, golden):
    """SURVEY 8 f1, generator side: StyleGan2Generator(blur_size=3) puts Blur(3) in the `# TODO: blur` slot
    (stylegan2.py:47); image and every parameter gradient against the oracle's composition in fp64."""
    from gan_amd.modelling import build_generator
    from oracle import gan_oracle as O
    from oracle.detparams import det_state, det_uniform
    from oracle.tape import RandTape

    gen = build_generator("stylegan2", img_size=16, base_dim=128, blur_size=3)
    keys = [k for k in gen.state_dict() if not k.endswith("kernel")]
    shapes = {k: tuple(gen.state_dict()[k].shape) for k in keys}
    assert "blocks.1.layers.3.scale" in shapes and "blocks.1.layers.5.weight" in shapes  # Blur sits at layers.2
    state = det_state(shapes, seed=41)
    gen.load_state_dict(state, strict=False)
    gen = gen.to(dev)
    z = det_uniform((4, 128), 431) * 3.0
    torch.manual_seed(4)
    with RandTape() as rec:
        img = gen(z.to(dev))
    wgt = det_uniform(tuple(img.shape), 432)
    ((img * wgt.to(dev)).sum() + img.square().mean()).backward()
    sd = {k: v.double().requires_grad_(True) for k, v in state.items()}
    img_o = O.generator(sd, z.double(), [t.double() for t in rec.tape], blur_size=3)
    ((img_o * wgt.double()).sum() + img_o.square().mean()).backward()
    assert rel_err(img, img_o) < TOL
    named = dict(gen.named_parameters())
    for k, v in sd.items():
        assert rel_err(named[k].grad, v.grad) < TOL, (k, rel_err(named[k].grad, v.grad))


def test_trainer_runs_stylegan_v1_recipe(dev):
    """README.md:59 recipe with --model stylegan (v1 generator + the shared discriminator, hinge + R1, bf16): three
    iterations through the Trainer, graph replay included, finite losses."""
    from gan_amd.train_celeba import TrainConfig, Trainer

    torch.manual_seed(0)
    cfg = TrainConfig(model="stylegan", img_size=32, gen_kwargs={"base_dim": 32}, disc_kwargs={"base_dim": 16}, method="hinge", regularizer="r1",
                      batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0.0, 0.9]}, lr=2e-4, device="cuda", mixed_precision=True,
                      channels_last=True, hip_graph=True, graph_warmup=1)
    tr = Trainer(cfg, dev)
    reals = torch.randn(8, 3, 32, 32, device=dev)
    for _ in range(4):
        logs = tr.train_step(lambda: reals.clone())
    assert tr._graph is not None
    assert all(bool(torch.isfinite(v).all()) for v in logs.values()), logs


@pytest.mark.gpu
def test_generator_bulk_noise_matches_per_layer_draws(monkeypatch):
    """StyleGan2Generator.bulk_noise: the Noise layers take consecutive slices of one draw.  Fed the same values layer by
    layer (torch.randn patched to hand out those slices), the per-layer path produces the same image bit for bit; the
    slices are exactly the (N, 1, H, W) draws the reference makes (stylegan.py:26), in its order."""
    import gan_amd.functional as GF
    from gan_amd.modelling import build_generator

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    gen = build_generator("stylegan2", img_size=32, base_dim=64).to(dev)
    for m in gen.modules():
        if type(m).__name__ == "Noise":
            torch.nn.init.normal_(m.scale, std=0.5)  # (zero at init: the draws would not matter)
    z = torch.randn(4, 128, device=dev)
    real_randn = torch.randn
    drawn = []

    def spy(*shape, **kw):
        out = real_randn(*shape, **kw)
        drawn.append(out)
        return out

    monkeypatch.setattr(torch, "randn", spy)
    gen.bulk_noise = True
    with torch.no_grad():
        img_bulk = gen(z)
    assert len(drawn) == 1 and drawn[0].dim() == 1  # one launch for the whole pass
    pool, cursor, shapes = drawn[0], [0], []

    def replay(*shape, **kw):
        n, _, h, w = shape
        size = n * h * w
        out = pool[cursor[0] : cursor[0] + size].view(n, 1, h, w).to(kw.get("dtype") or pool.dtype)
        cursor[0] += size + ((-size) % 8)
        shapes.append((h, w))
        return out

    monkeypatch.setattr(torch, "randn", replay)
    gen.bulk_noise = False
    with torch.no_grad():
        img_layers = gen(z)
    monkeypatch.setattr(torch, "randn", real_randn)
    assert shapes == [(4, 4), (8, 8), (8, 8), (16, 16), (16, 16), (32, 32), (32, 32)]
    assert torch.equal(img_bulk, img_layers)


def test_main_writes_sample_grids_and_checkpoints(dev, tmp_path, monkeypatch):
    """gan_amd.train_celeba.main: the 10 x 10 grids of the fixed latents for gen and gen_ema every log_img_interval
    iterations (train_celeba.py:162-168: images_celeba/<run_name>/step0000k[_ema].png), a resumable checkpoint every
    checkpoint_interval, and --resume continuing from it."""
    import struct

    from gan_amd import train_celeba as T

    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    argv = ["--model", "dcgan", "--img_size", "64", "--disc_kwargs", '{"norm": "none"}', "--method", "hinge", "--batch_size", "8", "--n_iters", "4",
            "--log_img_interval", "2", "--checkpoint_interval", "4", "--run_name", "t", "--optimizer_kwargs", '{"betas": [0.0, 0.9]}', "--device", "cuda"]
    T.main(argv)
    d = tmp_path / "images_celeba" / "t"
    for name in ("step0000k.png", "step0000k_ema.png"):
        raw = (d / name).read_bytes()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        w, h = struct.unpack(">II", raw[16:24])
        assert (w, h) == (640, 640)
    ck = d / "step0000004.pt"
    state = torch.load(ck, map_location="cpu", weights_only=True)
    assert state["step"] == 4
    T.main(argv[:-2] + ["--device", "cuda", "--resume", str(ck), "--n_iters", "6"])  # (argparse: the later --n_iters wins)
