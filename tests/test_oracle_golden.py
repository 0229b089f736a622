"""CPU: the oracle (oracle/gan_oracle.py) against the vectors the reference itself produced
(tests/golden/*.pt, written by oracle/make_goldens.py).  fp32 tolerance: rel-err <= 1e-5
(north_star); second-order quantities and 10-step trajectories get 1e-4."""
import math

import pytest
import torch

from conftest import rel_err
from oracle import gan_oracle as O
from oracle.detparams import det_state, det_uniform, projections, recipe_state
from oracle.tape import RandTape

TOL = 1e-5
TOL2 = 1e-4


def leaves_of(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


@pytest.mark.parametrize("tag,demod", [("k3_demod", True), ("k1_plain", False)])
def test_modconv(golden, tag, demod):
    g = golden("modconv")[tag]
    sd = leaves_of(g["state"])
    x = g["x"].clone().requires_grad_(True)
    w = g["w"].clone().requires_grad_(True)
    y = O.modulated_conv2d(x, w, sd["weight"], sd["bias"], sd["style_weight.weight"], sd["style_weight.bias"], demod)
    assert rel_err(y, g["y"]) < TOL
    names = ["x", "w", "weight", "bias", "style_weight.weight", "style_weight.bias"]
    leaves = [x, w] + [sd[n] for n in names[2:]]
    g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < TOL, n
    g2 = torch.autograd.grad(g1[0].square().sum() + g1[1].square().sum(), leaves, allow_unused=True)
    for n, t in zip(names, g2):
        ref = g["grad2"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert t is None or float(t.abs().max()) < 1e-6, n
        else:
            assert rel_err(t, ref) < TOL2, n


def test_upfirdn2d_cases(golden):
    for case in golden("upfirdn2d")["cases"]:
        x = case["x"].clone().requires_grad_(True)
        y = O.upfirdn2d(x, case["kernel"], *case["args"])
        assert y.shape == case["y"].shape
        assert rel_err(y, case["y"]) < TOL, case["args"]
        (g1,) = torch.autograd.grad(y.square().sum(), x, create_graph=True)
        assert rel_err(g1, case["g1"]) < TOL, case["args"]
        (g2,) = torch.autograd.grad(g1.square().sum(), x)
        assert rel_err(g2, case["g2"]) < TOL2, case["args"]


def test_blur_kernels_and_identities(golden):
    g = golden("upfirdn2d")
    for k, ref in g["make_kernel"].items():
        assert torch.equal(O.blur_kernel(int(k)), ref)
    with pytest.raises(AssertionError):
        O.blur_kernel(8)
    with pytest.raises(AssertionError):
        O.blur_kernel(1)
    x = g["identities"]["x"]
    assert torch.equal(O.nearest_up2(x), g["identities"]["nearest_up2"])
    assert rel_err(O.avg_pool2(x), g["identities"]["avg_pool2"]) < 1e-6
    # both are upfirdn2d special cases
    assert torch.equal(O.upfirdn2d(x, torch.ones(2, 2), 2, 1, 1, 0, 1, 0), g["identities"]["nearest_up2"])
    assert rel_err(O.upfirdn2d(x, torch.ones(2, 2) / 4, 1, 2, 0, 0, 0, 0), g["identities"]["avg_pool2"]) < 1e-6
    xb = g["blur"]["x"]
    for key, ref in g["blur"].items():
        if key == "x":
            continue
        k, u, d = (int(p[1:]) for p in key.split("_"))
        assert rel_err(O.blur(xb, k, u, d), ref) < TOL, key


def test_noise_mbstd_bias_lrelu(golden):
    g = golden("small_ops")
    n = g["noise"]
    assert rel_err(O.add_noise(n["x"], n["scale"], n["draw"]), n["y"]) < TOL
    with RandTape([n["draw"]]):
        assert rel_err(O.add_noise(n["x"], n["scale"]), n["y"]) < TOL
    m = g["mbstd"]
    x = m["x"].clone().requires_grad_(True)
    y = O.minibatch_stddev(x)
    assert rel_err(y, m["y"]) < TOL
    (g1,) = torch.autograd.grad((y * m["wgt"]).sum() + y[:, -1].square().sum(), x, create_graph=True)
    assert rel_err(g1, m["g1"]) < TOL
    (g2,) = torch.autograd.grad(g1.square().sum(), x)
    assert rel_err(g2, m["g2"]) < TOL2
    b = g["bias_lrelu"]
    x = b["x"].clone().requires_grad_(True)
    bias = b["b"].clone().requires_grad_(True)
    y = O.bias_leaky_relu(x, bias)
    assert rel_err(y, b["y"]) < TOL
    gx, gb = torch.autograd.grad(y.square().sum(), [x, bias], create_graph=True)
    assert rel_err(gx, b["gx"]) < TOL and rel_err(gb, b["gb"]) < TOL
    ggx, ggb = torch.autograd.grad(gx.square().sum() + gb.square().sum(), [x, bias])
    assert rel_err(ggx, b["ggx"]) < TOL2 and rel_err(ggb, b["ggb"]) < TOL2


def test_losses_and_ema(golden):
    g = golden("small_ops")
    lo = g["losses"]
    for name in ("gan", "lsgan", "wgan", "hinge", "relativistic-gan"):
        d, gl = lo[name]
        assert math.isclose(float(O.d_loss(name, lo["d_reals"], lo["d_fakes"])), d, rel_tol=1e-6, abs_tol=1e-7)
        assert math.isclose(float(O.g_loss(name, lo["d_fakes"], lambda: lo["d_reals"])), gl, rel_tol=1e-6, abs_tol=1e-7)
    with pytest.raises(KeyError):
        O.d_loss("nope", lo["d_reals"], lo["d_fakes"])
    assert O.regularizer("none") is None
    with pytest.raises(KeyError):
        O.regularizer("r2")
    e = g["ema"]
    model = {k: v.clone() for k, v in e["init"].items()}
    ema = {k: v.clone() for k, v in e["init"].items()}
    counter = 0
    for step in range(1, 104):
        model["weight"] += e["drift"]
        model["bias"] += 0.001 * step
        counter = O.ema_step(list(ema.values()), list(model.values()), counter, beta=0.999, warmup=100)
        if str(step) in e:
            snap = e[str(step)]
            assert counter == snap["counter"]
            for k in ema:
                assert rel_err(ema[k], snap["ema"][k]) < 1e-6, (step, k)


def test_disc_block(golden):
    g = golden("disc")["block"]
    sd = leaves_of(g["state"])
    x = g["x"].clone().requires_grad_(True)
    y = O.disc_block(sd, "", x)
    assert rel_err(y, g["y"]) < TOL
    names = list(g["grad"].keys())
    leaves = [x] + [sd[n] for n in names[1:]]
    g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
    for n, t in zip(names, g1):
        assert rel_err(t, g["grad"][n]) < TOL, n
    g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
    for n, t in zip(names, g2):
        ref = g["grad2"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert t is None or float(t.abs().max()) < 1e-6, n
        else:
            assert rel_err(t, ref) < TOL2, n


def _check_grads(sd, ref_grads, tol):
    """Per-tensor rel-err; tensors whose reference gradient is round-off noise (mathematically zero,
    e.g. conv biases under R1) are checked against the largest gradient's scale instead."""
    scale = max(float(r.abs().max()) for r in ref_grads.values() if r is not None)
    for k, ref in ref_grads.items():
        got = sd[k].grad
        if ref is None or float(ref.abs().max()) < 1e-7 * scale:
            assert got is None or float(got.abs().max()) < 1e-5 * scale, k
        else:
            assert got is not None, k
            assert rel_err(got, ref) < tol, (k, rel_err(got, ref))


def _zero(sd):
    for v in sd.values():
        v.grad = None


def test_disc16_hinge_r1_gp(golden):
    g = golden("disc")["disc16"]
    sd = leaves_of(g["state"])
    disc = lambda imgs: O.discriminator(sd, imgs)  # noqa: E731
    reals, fakes = g["reals"], g["fakes"]
    assert rel_err(disc(reals), g["logits_reals"]) < TOL

    loss = O.d_loss("hinge", disc(reals), disc(fakes))
    loss.backward()
    assert math.isclose(float(loss), g["hinge"]["value"], rel_tol=1e-5)
    _check_grads(sd, g["hinge"]["grads"], TOL2)

    _zero(sd)
    r = reals.clone().requires_grad_()
    r1 = O.r1_penalty(disc, r, fakes, disc(r))
    r1.backward()
    assert math.isclose(float(r1), g["r1"]["value"], rel_tol=1e-5)
    _check_grads(sd, g["r1"]["grads"], TOL2)
    assert rel_err(r.grad, g["r1"]["grad_reals"]) < TOL2

    _zero(sd)
    r = reals.clone().requires_grad_()
    d_r = disc(r)
    total = O.d_loss("hinge", d_r, disc(fakes)) + O.r1_penalty(disc, r, fakes, d_r)
    total.backward()
    assert math.isclose(float(total), g["hinge_r1"]["value"], rel_tol=1e-5)
    _check_grads(sd, g["hinge_r1"]["grads"], TOL2)

    _zero(sd)
    gp = O.wgan_gp_penalty(disc, reals, fakes, None, alpha=g["wgan_gp"]["alpha"])
    gp.backward()
    assert math.isclose(float(gp), g["wgan_gp"]["value"], rel_tol=1e-5)
    _check_grads(sd, g["wgan_gp"]["grads"], TOL2)


def test_generator16(golden):
    g = golden("generator16")
    shapes = O.stylegan2_generator_shapes(g["img_size"], base_dim=g["base_dim"])
    sd = leaves_of(det_state(shapes, seed=g["seed"]))
    img = O.generator(sd, g["z"], g["noise"])
    assert rel_err(img, g["img"]) < TOL
    loss = (img * det_uniform(tuple(img.shape), 902)).sum() + img.square().mean()
    assert math.isclose(float(loss), g["loss"], rel_tol=1e-5)
    loss.backward()
    for k, ref in g["grads"].items():
        got = sd[k].grad
        if sd[k].numel() > 4096:
            got = projections(got)
            # projections are sums of ~1e5..1e6 signed terms: compare against the abs-sum scale
            scale = float(ref[1])
            assert float((got - ref).abs().max()) < 2e-5 * scale, k
        else:
            assert rel_err(got, ref) < TOL2, (k, rel_err(got, ref))


def test_dcgan_wgan_gp(golden):
    g = golden("dcgan")
    d_sd = leaves_of(g["d_state"])
    g_sd = leaves_of(g["g_state"])
    disc = lambda imgs: O.dcgan_discriminator(d_sd, imgs)  # noqa: E731
    fakes = O.dcgan_generator(g_sd, g["z"])
    assert rel_err(fakes, g["fakes"]) < TOL
    assert rel_err(disc(g["reals"]), g["logits_reals"]) < TOL
    gp = O.wgan_gp_penalty(disc, g["reals"], fakes.detach(), None, alpha=g["wgan_gp"]["alpha"])
    loss = O.d_loss("wgan", disc(g["reals"]), disc(fakes.detach())) + gp
    loss.backward()
    assert math.isclose(float(gp), g["wgan_gp"]["value"], rel_tol=1e-5)
    assert math.isclose(float(loss), g["wgan_gp"]["loss"], rel_tol=1e-5)
    _check_grads(d_sd, g["wgan_gp"]["grads"], TOL2)
    _zero(d_sd)
    gl = O.g_loss("wgan", disc(O.dcgan_generator(g_sd, g["z"])), None)
    gl.backward()
    assert math.isclose(float(gl), g["g_loss"]["value"], rel_tol=1e-5)
    _check_grads(g_sd, g["g_loss"]["grads"], TOL2)


def test_trajectory16(golden):
    g = golden("trajectory16")
    g_sd = leaves_of(det_state(O.stylegan2_generator_shapes(16, base_dim=128), seed=g["gen_seed"], bias_amp=0.0, scale_amp=0.05))
    d_sd = leaves_of(det_state(O.stylegan2_discriminator_shapes(16, base_dim=16), seed=g["disc_seed"], bias_amp=0.0, scale_amp=0.5))
    ema = [v.detach().clone() for v in g_sd.values()]
    counter = 0
    opt_d = torch.optim.Adam(list(d_sd.values()), 2e-4, betas=(0.0, 0.9))
    opt_g = torch.optim.Adam(list(g_sd.values()), 2e-4, betas=(0.0, 0.9))
    for step in range(10):
        reals = det_uniform((4, 3, 16, 16), g["reals_salt0"] + step) * 2.0
        with RandTape(g["tape"][step]):
            logs = O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals, method="hinge", reg="r1")
        counter = O.ema_step(ema, list(g_sd.values()), counter, warmup=g["ema_warmup"])
        ref = g["steps"][step]
        assert math.isclose(logs["loss_d"], float(ref[0]), rel_tol=1e-4, abs_tol=1e-5), (step, logs, ref)
        assert math.isclose(logs["loss_g"], float(ref[1]), rel_tol=1e-4, abs_tol=1e-5), (step, logs, ref)
    for k, ref in g["final"]["gen"].items():
        got = projections(g_sd[k])
        assert float((got - ref).abs().max()) < 1e-4 * max(float(ref[1]), 1e-3), k
    for (k, ref), e in zip(g["final"]["ema"].items(), ema):
        got = projections(e)
        assert float((got - ref).abs().max()) < 1e-4 * max(float(ref[1]), 1e-3), k


def _recipe_fns(model):
    return (O.dcgan_generator, O.dcgan_discriminator) if model == "dcgan" else (O.generator, O.discriminator)


def test_trainer_recipes(golden):
    """The oracle's train iteration on the README recipes (DCGAN WGAN-GP with n_disc = 5, DCGAN WGAN with weight clipping,
    StyleGAN2 64 x 64 / base_dim 64 hinge + R1) against the losses and final parameters of the REFERENCE's modules run
    through the train_celeba.py:111-149 loop (oracle/make_goldens.py: gold_trainer_recipes)."""
    for name, g in golden("trainer_recipes").items():
        gen_fn, disc_fn = _recipe_fns(g["model"])
        g_sd = leaves_of(recipe_state(g["gen_shapes"], g["gen_seed"], 0.05, g["wgain"]))
        d_sd = leaves_of(recipe_state(g["disc_shapes"], g["disc_seed"], 0.5, g["wgain"]))
        opt_d = torch.optim.Adam(list(d_sd.values()), g["lr"], betas=(0.0, 0.9))
        opt_g = torch.optim.Adam(list(g_sd.values()), g["lr"], betas=(0.0, 0.9))
        salt = [g["reals_salt0"]]

        def next_reals():
            salt[0] += 1
            return det_uniform((g["batch"], 3, g["img_size"], g["img_size"]), salt[0]) * 2.0

        for step, ref in enumerate(g["steps"]):
            with RandTape(g["tape"][step]):
                logs = O.train_iteration(g_sd, d_sd, opt_g, opt_d, next_reals, method=g["method"], reg=g["regularizer"], n_disc=g["n_disc"],
                                         gen_fn=gen_fn, disc_fn=disc_fn)
            got = logs["loss_d_all"] + [logs["loss_g"]]
            scale = max(1.0, float(ref.abs().max()))  # (a WGAN loss is a difference of terms of this size)
            for j, (a, b) in enumerate(zip(got, ref.tolist())):
                assert math.isclose(a, b, rel_tol=2e-4, abs_tol=2e-5 * scale), (name, step, j, a, b)
        for sd, key in ((g_sd, "gen"), (d_sd, "disc")):
            for k, ref in g["final"][key].items():
                if not torch.is_floating_point(sd[k]) if k in sd else True:
                    continue
                got = projections(sd[k])
                assert float((got - ref).abs().max()) < 2e-4 * max(float(ref[1]), 1e-3), (name, key, k)


def test_resample_convs(golden):
    """up_conv_blur / blur_conv_down (nvidia_ops.py:81-123), every branch, against the reference's outputs."""
    for tag, g in golden("resample_convs").items():
        fn = getattr(O, g["fn"])
        sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("kernel")) for k, v in g["state"].items()}
        x = g["x"].clone().requires_grad_(True)
        y = fn(sd, x, g["kernel_size"], g["blur_size"])
        assert y.shape == g["y"].shape, tag
        assert rel_err(y, g["y"]) < 1e-5, tag
        names = list(g["grad"].keys())
        leaves = [x] + [sd[n] for n in names[1:]]
        g1 = torch.autograd.grad(y.square().sum(), leaves, create_graph=True)
        for n, t in zip(names, g1):
            assert rel_err(t, g["grad"][n]) < 1e-5, (tag, n)
        g2 = torch.autograd.grad(g1[0].square().sum(), leaves, allow_unused=True)
        for n, t in zip(names, g2):
            ref = g["grad2"][n]
            if ref is None:
                assert t is None or float(t.abs().max()) == 0.0, (tag, n)
            else:
                assert rel_err(t, ref) < 1e-4, (tag, n)


@pytest.mark.parametrize("name", ["stylegan", "progressive_gan"])
def test_generators_v1(golden, name):
    """StyleGanGenerator / ProgressiveGanGenerator (stylegan.py:50-85, progressive_gan.py:103-133; AdaIN and LayerNorm2d,
    common.py:7-24) against the reference's image, loss and parameter gradients."""
    g = golden("generators_v1")[name]
    shapes = {k: tuple(v) for k, v in g["shapes"].items()}
    sd = leaves_of(det_state(shapes, seed=g["seed"], bias_amp=0.1, scale_amp=0.1))
    img = O.stylegan_generator(sd, g["z"], list(g["noise"])) if name == "stylegan" else O.progan_generator(sd, g["z"])
    assert rel_err(img, g["img"]) < TOL
    loss = (img * det_uniform(tuple(img.shape), 950 + g["seed"])).sum() + img.square().mean()
    assert math.isclose(float(loss), g["loss"], rel_tol=1e-5)
    loss.backward()
    for k, ref in g["grads"].items():
        got = sd[k].grad
        if sd[k].numel() > 4096:
            assert float((projections(got) - ref).abs().max()) < 2e-4 * float(ref[1]), k
        else:
            assert rel_err(got, ref) < TOL2, (k, rel_err(got, ref))


def test_diff_augment(golden):
    """oracle.diff_augment (draws explicit) against the reference's DiffAugment()(x) with its draws recovered by re-seeding
    (oracle/make_goldens.py: gold_diff_augment): image, gradient, gradient of the gradient's norm."""
    for tag, g in golden("diff_augment").items():
        x = g["x"].double().requires_grad_(True)
        tx, ty = g["translate"]
        oy, ox = g["cutout"]
        y = O.diff_augment(x, g["brightness"].double(), g["saturation"].double(), g["contrast"].double(), tx, ty, oy, ox)
        assert rel_err(y, g["y"]) < 1e-6, tag
        loss = (y * g["wgt"].double()).sum() + y.square().mean()
        assert abs(float(loss) - g["loss"]) <= 1e-5 * max(1.0, abs(g["loss"]))
        (g1,) = torch.autograd.grad(loss, x, create_graph=True)
        (g2,) = torch.autograd.grad(g1.square().sum(), x)
        assert rel_err(g1, g["grad"]) < 1e-5 and rel_err(g2, g["grad2"]) < 1e-5, tag
