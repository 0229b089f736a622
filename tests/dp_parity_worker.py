"""Worker of tests/test_zz_multirank.py::test_two_rank_trainer_matches_global_batch (not collected by pytest).

Launched with torch.distributed.run, 2 ranks sharing the one GPU (GAN_AMD_ONE_GPU=1: gloo, host-staged collectives).
Each rank runs the REAL Trainer on its interleaved shard of one global batch (global z / noise draws, sliced), both the
host-launched path (bucketed all-reduces from the hooks) and the cut-iteration path the HIP graphs capture (async D
all-reduces under the generator forward the next pass needs; n_disc = 2, grad_accum = 2 and the deferred generator
update as well, with the list of cuts asserted), and compares the averaged D and G gradient arenas with the gradients of the SAME
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
    pool = torch.randn(4, GLOBAL, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(3))

    real_randn = torch.randn
    gen_cpu = torch.Generator()
    crit, reg = get_loss("hinge"), get_regularizer("r1")
    snaps = {}
    for name, opt, red in (("d", tr.optim_d, tr.red_d), ("g", tr.optim_g, tr.red_g)):
        def step(name=name, red=red):
            snaps[name] = red.arena.grad.clone()  # lr = 0: nothing to update, the averaged gradients are what counts
        opt.step = step
    worst = 0.0
    # (n_disc, grad_accum, deferred generator update, the cuts the graph path must make)
    cases = ((1, 1, False, ["start", "wait", "reduce"]),
             (2, 1, False, ["start", "wait", "start", "wait", "reduce"]),
             (1, 2, False, ["start", "wait", "reduce"]),
             (2, 1, True, ["finish_g", "start", "wait", "start", "wait", "start_g"]))
    for n_disc, accum, defer, want_cuts in cases:
        tr.cfg.n_disc, tr.cfg.grad_accum = n_disc, accum
        mg = GLOBAL // accum              # a micro-batch of the whole job, and this rank's rows of it
        mrows = interleaved_shard(mg, rank, world)
        local = mg // world

        def make_randn(shard: bool):
            def randn(*shape, **kw):
                shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
                batchy = shape[0] == (local if shard else mg)
                t = real_randn((mg,) + shape[1:] if batchy else shape, generator=gen_cpu)
                if batchy and shard:
                    t = t[mrows]
                return t.to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)

            return randn

        def feeder(shard: bool):
            k = [0]

            def feed():
                batch = pool[k[0] % 4, :mg]
                k[0] += 1
                return (batch[mrows] if shard else batch).to(dev)

            return feed

        # ---- reference: the iteration on the global (micro-)batches, gradients summed over grad_accum ------------
        gen_cpu.manual_seed(77)
        torch.randn = make_randn(False)
        try:
            feed = feeder(False)
            for i in range(n_disc):  # lr = 0: every D step sees the same weights; the last one's gradients are compared
                want_d = None
                for a in range(accum):
                    reals = feed().requires_grad_(True)
                    zs = torch.randn(mg, 128, device=dev)
                    with torch.no_grad():
                        fakes = ref_g(zs)
                    d_r, d_f = ref_d(reals), ref_d(fakes)
                    loss_d = crit.d_loss(d_r, d_f) + reg(ref_d, reals, fakes, d_r)
                    gr = torch.autograd.grad(loss_d, list(ref_d.parameters()))
                    want_d = gr if want_d is None else [x + y for x, y in zip(want_d, gr)]
            ref_d.requires_grad_(False)
            want_g = None
            for a in range(accum):
                zs = torch.randn(mg, 128, device=dev)
                loss_g = crit.g_loss(ref_d(ref_g(zs)), None)
                gr = torch.autograd.grad(loss_g, list(ref_g.parameters()))
                want_g = gr if want_g is None else [x + y for x, y in zip(want_g, gr)]
            ref_d.requires_grad_(True)
        finally:
            torch.randn = real_randn

        # ---- the data-parallel trainer on this rank's shard, both launch paths ----------------------------------
        for path in ("hooks", "cuts"):
            if defer and path == "hooks":
                continue  # (the deferred update exists on the graph path only)
            gen_cpu.manual_seed(77)
            torch.randn = make_randn(True)
            snaps.clear()
            try:
                feed = feeder(True)
                tr.red_d.zero_grad(); tr.red_g.zero_grad()
                if path == "hooks":
                    logs = tr._run_eager(feed)
                else:
                    it = tr._iteration(feed, overlap=False, defer_g=defer)
                    cuts = []
                    try:
                        while True:
                            cut = next(it)
                            cuts.append(cut[0])
                            tr._exchange(*cut)
                    except StopIteration as stop:
                        logs = stop.value
                    assert cuts == want_cuts, (cuts, want_cuts)
                    if defer:
                        assert tr._pending_g and "g" not in snaps  # the averaged gradients wait for the next iteration
                        tr.flush()
                        assert not tr._pending_g
            finally:
                torch.randn = real_randn
            for name, want, red in (("d", want_d, tr.red_d), ("g", want_g, tr.red_g)):
                got = snaps[name]
                for p, o, w in zip(red.arena.params, red.arena.offsets, want):
                    e = rel(got[o:o + p.numel()].view(p.shape), w)
                    worst = max(worst, e)
                    assert e < 2e-4, (n_disc, accum, defer, path, name, tuple(p.shape), e)
            # the logged losses are local means (last pass): their average over ranks is the global-batch loss
            t = torch.stack([logs["loss/d"].float(), logs["loss/g"].float()]).cpu()
            dist.all_reduce(t)
            t /= world
            assert abs(float(t[0]) - float(loss_d)) < 1e-4 * max(1.0, abs(float(loss_d))), (path, t, loss_d)
            assert abs(float(t[1]) - float(loss_g)) < 1e-4 * max(1.0, abs(float(loss_g))), (path, t, loss_g)
            assert logs["loss/d_steps"].numel() == n_disc * accum
            # and every rank ended with the same averaged arenas
            for name in ("d", "g"):
                mine = snaps[name].cpu()
                other = mine.clone()
                dist.broadcast(other, src=0)
                assert torch.equal(mine, other)
    dist.barrier()
    if rank == 0:
        print(f"DP_PARITY_OK worst rel-err {worst:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
