"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the module
API mirrors the reference (names, state_dict keys, registry errors, CLI parser), the product refuses to
compute without a GPU, the flat arenas, and the data-parallel plumbing on gloo with world_size 2."""
import os
import re
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from conftest import ROOT, rel_err


def test_library_exports_every_declared_symbol():
    from gan_amd import _lib

    header = open(os.path.join(ROOT, "include", "gan_amd.h")).read()
    declared = set(re.findall(r"\b(gan_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    lib = _lib.load()  # raises if the .so is missing or a symbol is absent
    assert lib.gan_amd_version() == _lib.ABI_VERSION
    assert lib.gan_amd_arch() == b"gfx950"
    for name in declared:
        assert hasattr(lib, name)


def test_no_cpu_fallback():
    from gan_amd import _lib
    from gan_amd.modelling import build_discriminator, build_generator
    from gan_amd.modelling.nvidia_ops import upfirdn2d

    with pytest.raises(_lib.GanAmdError):
        upfirdn2d(torch.randn(1, 8, 4, 4), torch.ones(3, 3), 1, 1, 1, 1, 1, 1)
    with pytest.raises(_lib.GanAmdError):
        build_discriminator("stylegan2", img_size=16, base_dim=8)(torch.randn(4, 3, 16, 16))
    with pytest.raises(_lib.GanAmdError):
        build_generator("stylegan2", img_size=16, base_dim=128)(torch.randn(2, 128))


def test_product_does_not_import_oracle():
    import subprocess
    import sys

    code = "import sys; import gan_amd, gan_amd.train_celeba; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gan_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_registry_and_state_dict_parity():
    from gan_amd.modelling import build_discriminator, build_generator
    from oracle import gan_oracle as O

    g = build_generator("stylegan2", img_size=32, base_dim=64)
    d = build_discriminator("stylegan2", img_size=32, base_dim=16)
    assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == O.stylegan2_generator_shapes(32, base_dim=64)
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == O.stylegan2_discriminator_shapes(32, base_dim=16)
    # reference init: learned_input ones, Noise.scale zeros, D scale 1e-4, zero conv biases
    assert torch.all(g.learned_input == 1)
    assert all(torch.all(v == 0) for k, v in g.state_dict().items() if k.endswith(".scale"))
    assert all(torch.allclose(v, torch.full_like(v, 1e-4)) for k, v in d.state_dict().items() if k.endswith("scale"))
    assert all(torch.all(v == 0) for k, v in d.state_dict().items() if k.endswith("bias"))
    with pytest.raises(KeyError):
        build_generator("nope", img_size=32)
    with pytest.raises(KeyError):
        build_discriminator("nope", img_size=32)
    # first block is hard-wired to 512 channels: 64x64 with base_dim 16 is inconsistent, like the reference
    dc_d = build_discriminator("dcgan", img_size=64, base_dim=8, norm="none")
    dc_g = build_generator("dcgan", img_size=64, base_dim=8)
    assert "0.0.weight" in dc_d.state_dict() and "4.weight" in dc_d.state_dict()
    assert "0.1.weight" in dc_g.state_dict() and "0.2.weight" in dc_g.state_dict()


def test_dcgan_state_dict_matches_golden_keys(golden):
    from gan_amd.modelling import build_discriminator, build_generator

    g = golden("dcgan")
    dc_d = build_discriminator("dcgan", img_size=64, base_dim=8, norm="none")
    dc_g = build_generator("dcgan", img_size=64, base_dim=8)
    assert {k: tuple(v.shape) for k, v in dc_d.state_dict().items()} == {k: tuple(v.shape) for k, v in g["d_state"].items()}
    assert {k: tuple(v.shape) for k, v in dc_g.state_dict().items()} == {k: tuple(v.shape) for k, v in g["g_state"].items()}


def test_make_parser_and_losses_registry():
    from gan_amd.losses import get_loss, get_regularizer
    from gan_amd.train_celeba import TrainConfig
    from gan_amd.utils import make_parser

    args = make_parser(TrainConfig).parse_args(
        ["--model", "stylegan2", "--img_size", "64", "--gen_kwargs", '{"base_dim": 64}', "--mixed_precision", "--optimizer_kwargs", '{"betas": [0, 0.9]}', "--lr", "1e-4"]
    )
    cfg = TrainConfig(**vars(args))
    assert cfg.model == "stylegan2" and cfg.img_size == 64 and cfg.gen_kwargs == {"base_dim": 64}
    assert cfg.mixed_precision is True and cfg.channels_last is False and cfg.lr == 1e-4 and cfg.n_disc == 1

    def bad(x: list = None):
        pass

    with pytest.raises(RuntimeError):
        make_parser(bad)
    for name in ("gan", "lsgan", "wgan", "hinge", "relativistic-gan"):
        assert hasattr(get_loss(name), "d_loss") and hasattr(get_loss(name), "g_loss")
    assert get_regularizer("none") is None and callable(get_regularizer("r1")) and callable(get_regularizer("wgan-gp"))
    with pytest.raises(KeyError):
        get_loss("nope")
    with pytest.raises(KeyError):
        get_regularizer("nope")


def test_losses_values_match_golden(golden):
    from gan_amd.losses import get_loss

    lo = golden("small_ops")["losses"]
    for name in ("gan", "lsgan", "wgan", "hinge", "relativistic-gan"):
        crit = get_loss(name)
        assert abs(float(crit.d_loss(lo["d_reals"], lo["d_fakes"])) - lo[name][0]) < 1e-6
        assert abs(float(crit.g_loss(lo["d_fakes"], lambda: lo["d_reals"])) - lo[name][1]) < 1e-6


def test_flat_arena():
    from gan_amd.utils import FlatArena, arena_of

    m = nn.Sequential(nn.Linear(5, 3), nn.Linear(3, 7))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = arena_of(m, with_grad=True)
    assert arena.intact() and arena.numel % 4 == 0 and all(o % 4 == 0 for o in arena.offsets)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    m(torch.randn(2, 5)).sum().backward()
    for p, o in zip(arena.params, arena.offsets):  # grads accumulated in place into the arena
        assert p.grad.data_ptr() == arena.grad.data_ptr() + 4 * o
        assert torch.equal(arena.grad[o : o + p.numel()].view(p.shape), p.grad)
    assert arena_of(m) is arena
    m.to(torch.float64)
    assert not arena.intact()
    with pytest.raises(ValueError):
        FlatArena([])


def test_interleaved_shard_preserves_mbstd_groups():
    from gan_amd.distributed import interleaved_shard
    from oracle import gan_oracle as O

    n, world = 32, 4
    x = torch.randn(n, 6, 4, 4)
    ref = O.minibatch_stddev(x)
    seen = []
    for r in range(world):
        idx = interleaved_shard(n, r, world)
        seen.append(idx)
        local = O.minibatch_stddev(x[idx])
        assert torch.allclose(local, ref[idx], atol=1e-6)
    assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(n))
    with pytest.raises(ValueError):
        interleaved_shard(30, 0, 4)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gan_amd.distributed import GradBucketReducer, init_from_env, interleaved_shard

    r, w, device = init_from_env("cpu")
    assert (r, w, device.type) == (rank, world, "cpu")
    torch.manual_seed(0)
    model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.LeakyReLU(0.2), nn.Conv2d(8, 8, 3, padding=1), nn.Flatten(), nn.Linear(8 * 16, 1))
    frozen = nn.Linear(4, 4)  # never receives a gradient: its bucket must still be reduced in lock-step
    both = nn.ModuleList([model, frozen])
    red = GradBucketReducer(both, bucket_bytes=256)  # tiny buckets -> several collectives
    assert len(red.buckets) > 2
    x = torch.randn(16, 3, 4, 4)
    idx = interleaved_shard(16, rank, world)
    for _ in range(2):  # two passes: arenas are reused, gradients re-zeroed
        red.zero_grad()
        red.prepare()
        model(x[idx]).square().mean().backward()
        red.finish()
    torch.save(red.arena.grad.clone(), os.path.join(out_dir, f"g{rank}.pt"))
    if rank == 0:
        for p in model.parameters():
            p.grad = None
        model(x).square().mean().backward()
        torch.save(torch.cat([p.grad.flatten() for p in model.parameters()]), os.path.join(out_dir, "ref.pt"))
        torch.save([(o, p.numel()) for p, o in zip(red.arena.params, red.arena.offsets)], os.path.join(out_dir, "layout.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_reducer_gloo_world2(tmp_path):
    world = 2
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / "g0.pt")
    g1 = torch.load(tmp_path / "g1.pt")
    ref = torch.load(tmp_path / "ref.pt")
    layout = torch.load(tmp_path / "layout.pt")
    assert torch.equal(g0, g1)  # every rank holds the same averaged gradient
    n_model = sum(1 for _ in range(6))  # 6 parameter tensors in `model`
    got = torch.cat([g0[o : o + n] for o, n in layout[:n_model]])
    assert rel_err(got, ref) < 1e-6  # mean of shard gradients == full-batch gradient
    tail = torch.cat([g0[o : o + n] for o, n in layout[n_model:]])
    assert float(tail.abs().max()) == 0.0


def test_need_follows_the_running_autograd_task():
    """functional._need: `ctx.needs_input_grad` says whether an input requires grad at all; which gradients THIS backward
    pass uses is the engine's knowledge.  Under autograd.grad(out, inputs=[x]) -- the first pass of R1 / WGAN-GP
    (losses.py:95-103) -- a weight's gradient is not wanted; under backward() and when the weight is asked for it is.
    Argument positions are mapped to gradient edges (non-tensor arguments have none)."""
    import gan_amd.functional as GF

    seen = []

    class Toy(GF.Function):
        @staticmethod
        def forward(ctx, x, k, w, b=None):
            ctx.save_for_backward(x, w)
            return x * w * k

        @staticmethod
        def backward(ctx, g):
            seen.append([GF._need(ctx, i) for i in range(4)])
            x, w = ctx.saved_tensors
            return g * w, None, g * x, None

    x0 = torch.randn(3, requires_grad=True)
    w = torch.randn(3, requires_grad=True)
    torch.autograd.grad(Toy.apply(x0 * 2, 2.0, w, None).sum(), x0, create_graph=True)
    Toy.apply(x0 * 2, 2.0, w, None).sum().backward()
    torch.autograd.grad(Toy.apply(x0 * 2, 2.0, w, None).sum(), [x0, w])
    torch.autograd.grad(Toy.apply(x0 * 2, 2.0, w, None).sum(), [w])
    Toy.apply(x0 * 2, 2.0, w, None).sum().backward(inputs=[w])
    assert seen == [[True, False, False, False], [True, False, True, False], [True, False, True, False], [False, False, True, False],
                    [False, False, True, False]]
