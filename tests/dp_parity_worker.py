"""Worker of tests/test_zz_multirank.py::test_two_rank_trainer_matches_global_batch (not collected by pytest).

Launched with torch.distributed.run, 2 ranks sharing the one GPU (GAN_AMD_ONE_GPU=1: gloo, host-staged collectives).
Each rank runs the REAL Trainer on its interleaved shard of one global batch (global z / noise draws, sliced), both the
host-launched path (bucketed all-reduces from the hooks) and the cut-iteration path the HIP graphs capture (async D
all-reduce under the G forward), and compares the averaged D and G gradient arenas with the gradients of the SAME
iteration computed on the whole global batch by plain modules with no reducers.  MinibatchStdDev couples samples
across the batch (progressive_gan.py:27): only the interleaved shard reproduces its groups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gan_amd.distributed import init_from_env, interleaved_shard  # noqa: E402
from gan_amd.losses import get_loss, get_regularizer  # noqa: E402
from gan_amd.modelling import build_discriminator, build_generator  # noqa: E402
from gan_amd.train_celeba import TrainConfig, Trainer  # noqa: E402

GLOBAL, SIZE = 16, 32


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    rank, world, dev = init_from_env("cuda")
    assert world == 2
    local = GLOBAL // world
    rows = interleaved_shard(GLOBAL, rank, world)
    cfg = TrainConfig(model="stylegan2", img_size=SIZE, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16}, method="hinge",
                      regularizer="r1", batch_size=GLOBAL, optimizer="Adam", optimizer_kwargs={"betas": [0.0, 0.9]}, lr=0.0, device="cuda")
    torch.manual_seed(5)
    tr = Trainer(cfg, dev, world)  # broadcasts rank 0's parameters
    with torch.no_grad():  # non-zero Noise.scale and biases, identical on both ranks
        g = torch.Generator().manual_seed(9)
        for name, p in list(tr.gen.named_parameters()) + list(tr.disc.named_parameters()):
            if name.endswith("scale") or name.endswith("bias"):
                p.add_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
    # the same networks without arenas / reducers, for the global-batch reference
    ref_g = build_generator("stylegan2", img_size=SIZE, base_dim=64).to(dev)
    ref_d = build_discriminator("stylegan2", img_size=SIZE, base_dim=16).to(dev)
    ref_g.load_state_dict(tr.gen.state_dict())
    ref_d.load_state_dict(tr.disc.state_dict())
    reals_global = torch.randn(GLOBAL, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(3))

    real_randn = torch.randn
    gen_cpu = torch.Generator()

    def make_randn(shard: bool):
        def randn(*shape, **kw):
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            batchy = shape[0] == (local if shard else GLOBAL)
            t = real_randn((GLOBAL,) + shape[1:] if batchy else shape, generator=gen_cpu)
            if batchy and shard:
                t = t[rows]
            return t.to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)

        return randn

    # ---- reference: the iteration on the global batch ------------------------------------------------------
    crit, reg = get_loss("hinge"), get_regularizer("r1")
    gen_cpu.manual_seed(77)
    torch.randn = make_randn(False)
    try:
        reals = reals_global.to(dev).requires_grad_(True)
        zs = torch.randn(GLOBAL, 128, device=dev)
        with torch.no_grad():
            fakes = ref_g(zs)
        d_r, d_f = ref_d(reals), ref_d(fakes)
        loss_d = crit.d_loss(d_r, d_f) + reg(ref_d, reals, fakes, d_r)
        want_d = torch.autograd.grad(loss_d, list(ref_d.parameters()))
        ref_d.requires_grad_(False)
        zs = torch.randn(GLOBAL, 128, device=dev)
        loss_g = crit.g_loss(ref_d(ref_g(zs)), None)
        want_g = torch.autograd.grad(loss_g, list(ref_g.parameters()))
    finally:
        torch.randn = real_randn

    # ---- the data-parallel trainer on this rank's shard, both launch paths ----------------------------------
    snaps = {}
    for name, opt, red in (("d", tr.optim_d, tr.red_d), ("g", tr.optim_g, tr.red_g)):
        def step(name=name, red=red):
            snaps[name] = red.arena.grad.clone()  # lr = 0: nothing to update, the averaged gradients are what counts
        opt.step = step
    worst = 0.0
    for path in ("hooks", "cuts"):
        gen_cpu.manual_seed(77)
        torch.randn = make_randn(True)
        try:
            feed = lambda: reals_global[rows].to(dev)  # noqa: E731
            if path == "hooks":
                logs = tr._run_eager(feed)
            else:
                tr.red_d.zero_grad(); tr.red_g.zero_grad()
                it = tr._iteration(feed, overlap=False)
                cuts = []
                try:
                    while True:
                        cut = next(it)
                        cuts.append(cut[0])
                        tr._exchange(*cut)
                except StopIteration as stop:
                    logs = stop.value
                assert cuts == ["start", "wait", "reduce"], cuts
        finally:
            torch.randn = real_randn
        for name, want, red in (("d", want_d, tr.red_d), ("g", want_g, tr.red_g)):
            got = snaps[name]
            for p, o, w in zip(red.arena.params, red.arena.offsets, want):
                e = rel(got[o:o + p.numel()].view(p.shape), w)
                worst = max(worst, e)
                assert e < 2e-4, (path, name, tuple(p.shape), e)
        # the logged losses are local means: their average over ranks is the global-batch loss
        t = torch.stack([logs["loss/d"].float(), logs["loss/g"].float()]).cpu()
        dist.all_reduce(t)
        t /= world
        assert abs(float(t[0]) - float(loss_d)) < 1e-4 * max(1.0, abs(float(loss_d))), (path, t, loss_d)
        assert abs(float(t[1]) - float(loss_g)) < 1e-4 * max(1.0, abs(float(loss_g))), (path, t, loss_g)
    # and every rank ended with the same averaged arenas
    for red in (tr.red_d, tr.red_g):
        mine = snaps["d" if red is tr.red_d else "g"].cpu()
        other = mine.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(mine, other)
    dist.barrier()
    if rank == 0:
        print(f"DP_PARITY_OK worst rel-err {worst:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
