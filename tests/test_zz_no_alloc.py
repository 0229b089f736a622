"""gan_conv2d_wgrad_static -- the entry point that allocates nothing (include/gan_amd.h) -- against
gan_conv2d_wgrad with a caller-owned workspace, straight through the C ABI (ctypes), on two calls in a row, the
second with a smaller input.  Both entry points run the same kernels with the same split-K plan, so the
deterministic kernel paths (thin / tiled: plain stores into slices) must agree bit for bit; the register-staged
fallback accumulates with fp32 atomics and is held to 1e-5 (fp32) / 1e-2 (bf16 operands) relative error.

Collected last (file name): the round-end run uses `pytest -x`."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gan_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _operands(dev, dtype, n, cin, cout, h, w, k, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cout, h, w, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
    return x, gy


def _call(lib, static, x, gy, k, dw):
    from gan_amd._lib import check, dtype_code, ptr, stream_of

    n, cin, h, w = x.shape
    cout = gy.shape[1]
    pad = (k - 1) // 2
    dev = x.device
    shape_args = (n, h, w, cin, 0, h, w, cout, k, k, pad, pad)
    if static:
        check(lib.gan_conv2d_wgrad_static(ptr(x), ptr(gy), ptr(dw), cout, cin, None, *shape_args, 1.0, 0, 1.0, dtype_code(x), stream_of(x)),
              "gan_conv2d_wgrad_static")
    else:
        need = lib.gan_conv2d_wgrad_workspace(*shape_args, dtype_code(x))
        ws = torch.empty((max(need, 1),), dtype=torch.float32, device=dev)
        check(lib.gan_conv2d_wgrad(ptr(x), ptr(gy), ptr(dw), cout, cin, None, *shape_args, ptr(ws), need, 1.0, 0, 1.0, 0, dtype_code(x), stream_of(x)),
              "gan_conv2d_wgrad")
    return dw


# (dtype, first call (n, cin, cout, h, w, k), second -- smaller -- call, bit-exact?)
CASES = [
    # thin kernel (slices by plain stores), then a smaller thin shape
    (torch.bfloat16, (4, 16, 32, 64, 64, 3), (2, 16, 16, 32, 32, 3), True),
    # tiled kernel: 128 -> 128 on 16x16, then 64 -> 64 on 8x8
    (torch.bfloat16, (8, 128, 128, 16, 16, 3), (4, 64, 64, 8, 8, 3), True),
    # tiled kernel at the largest weight of the 256x256 networks (512 x 512 x 3 x 3, 4 slices = 37.7 MB of scratch), then 1x1
    (torch.bfloat16, (8, 512, 512, 4, 4, 3), (4, 256, 512, 4, 4, 1), True),
    # register-staged fallback (fp32: one atomically accumulated slice, cleared by the call), larger then smaller
    (torch.float32, (3, 32, 48, 12, 20, 3), (2, 16, 16, 5, 7, 3), False),
    # fallback after a tiled call left the scratch full of partial sums: the clear is what makes this right
    (torch.bfloat16, (8, 512, 512, 4, 4, 3), (2, 48, 16, 6, 10, 3), False),
]


@pytest.mark.parametrize("case", CASES)
def test_wgrad_static_matches_workspace_entry(dev, case):
    from gan_amd import _lib

    lib = _lib.load()
    assert lib.gan_amd_init() == 0
    dtype, first, second, exact = case
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for seed, (n, cin, cout, h, w, k) in enumerate((first, second)):
        x, gy = _operands(dev, dtype, n, cin, cout, h, w, k, seed)
        want = _call(lib, False, x, gy, k, torch.full((cout, cin, k, k), float("nan"), dtype=torch.float32, device=dev))
        got = torch.full((cout, cin, k, k), float("nan"), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]  # device-level (hipMemGetInfo), not torch's own bookkeeping
        _call(lib, True, x, gy, k, got)
        torch.cuda.synchronize()
        free_after = torch.cuda.mem_get_info(dev)[0]
        assert free_after == free_before, "the static entry point must not allocate device memory"
        assert not torch.isnan(got).any()
        if exact:
            assert torch.equal(got, want), (first, second, seed)
        else:
            assert rel_err(got, want) < tol, (first, second, seed)
        # and against the plain definition of the weight gradient (fp64 on the host), so that both being wrong shows
        xd, gd = x.double().cpu(), gy.double().cpu()
        ref = torch.nn.grad.conv2d_weight(xd, (cout, cin, k, k), gd, padding=(k - 1) // 2)
        assert rel_err(got, ref) < tol


def test_wgrad_static_rejects_oversize(dev):
    from gan_amd import _lib

    lib = _lib.load()
    cap = lib.gan_conv2d_wgrad_static_max_floats()
    assert cap == 12582912
    # fp32 path: one padded slice kh*kw*Cout*Cin = 9 * 2048 * 1024 floats > cap
    n, cin, cout, h, w, k = 1, 1024, 2048, 4, 4, 3
    assert lib.gan_conv2d_wgrad_workspace(n, h, w, cin, 0, h, w, cout, k, k, 1, 1, _lib.GAN_F32) > cap
    x = torch.zeros(16, device=dev)
    dw = torch.zeros(16, device=dev)
    rc = lib.gan_conv2d_wgrad_static(x.data_ptr(), x.data_ptr(), dw.data_ptr(), cout, cin, None, n, h, w, cin, 0, h, w, cout, k, k, 1, 1,
                                     1.0, 0, 1.0, _lib.GAN_F32, torch.cuda.current_stream().cuda_stream)
    assert rc == -1
    assert b"scratch" in lib.gan_amd_last_error()
    torch.cuda.synchronize()
    assert float(dw.abs().sum()) == 0.0  # nothing was launched
