"""The train_celeba.py step on the HIP kernels (reference: train_celeba.py:23-49 config, :60-77 set-up,
:111-149 loop body), with synthetic CelebA-shaped data and optional batch data parallelism.

    python -m gan_amd.train_celeba --model stylegan2 --img_size 64 --gen_kwargs '{"base_dim":64}' \
        --disc_kwargs '{"base_dim":64}' --method hinge --regularizer r1 --mixed_precision --channels_last \
        --optimizer Adam --optimizer_kwargs '{"betas":[0.0,0.9]}' --batch_size 64 --n_iters 100
    torchrun --nproc-per-node 8 -m gan_amd.train_celeba ... --batch_size 256     # global batch, sharded

Every `log_img_interval` iterations rank 0 writes the 10 x 10 grid of the fixed latents for `gen` and `gen_ema`
(train_celeba.py:162-168) under images_celeba/<run_name>/; `checkpoint_interval` > 0 saves a resumable checkpoint there
and `--resume PATH` continues from one.  FID (train_celeba.py:170-179) needs an Inception feature network and the CelebA
statistics, neither of which exists in this image: `run_fid(trainer, fid_scorer, ref_stats)` is the hook a caller with
both passes to main's loop through `fid=`; without it the interval is skipped.
Out of scope here: the CelebA loader, wandb.
"""
import json
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterator, List, Optional

import torch
from torch import Tensor, nn

from . import functional as GF
from .distributed import GradBucketReducer, all_reduce_sum, broadcast, init_from_env, interleaved_shard
from .losses import get_loss, get_regularizer
from .modelling import build_discriminator, build_generator
from .optim import make_optimizer
from .utils import EMA, make_parser, prepare_model


@dataclass
class TrainConfig:
    model: str = "dcgan"
    img_size: int = 64
    disc_kwargs: dict = field(default_factory=dict)
    gen_kwargs: dict = field(default_factory=dict)
    sn_disc: bool = False
    sn_gen: bool = False

    device: str = "cuda" if torch.cuda.is_available() else "cpu"
    mixed_precision: bool = False
    channels_last: bool = False
    compile: bool = False
    grad_accum: int = 1
    n_iters: int = 30_000
    n_disc: int = 1
    lr: float = 2e-4
    optimizer: str = "Adam"
    optimizer_kwargs: dict = field(default_factory=dict)
    batch_size: int = 64
    method: str = "gan"
    regularizer: str = "none"
    diff_augment: bool = False

    run_name: str = "dcgan_celeba"
    log_img_interval: int = 1_000
    fid_interval: int = 5_000

    # MI355X-side knobs (no counterpart in the reference's TrainConfig): replay the iteration as one HIP graph
    hip_graph: bool = False
    graph_warmup: int = 3
    # all Noise draws of a generator pass from one torch.randn launch (the same distribution, another walk through the
    # random stream than the reference's one draw per layer, stylegan.py:26)
    bulk_noise: bool = False
    # HIP-graph path: leave the generator's gradient all-reduce in flight at the end of an iteration and apply its
    # optimizer step + EMA update early in the next one, under work that does not read the generator (Trainer.flush()
    # settles it).  On whenever the job has more than one rank; this switch turns it on for a single rank too (tests).
    defer_g_step: bool = False
    # checkpoints (absent in the reference): every N iterations under images_celeba/<run_name>/ (0: never); resume path
    checkpoint_interval: int = 0
    resume: str = ""


def _fix_betas(kwargs: dict) -> dict:
    """README.md:60 passes betas [0, 0.9]; recent torch.optim.Adam rejects the int 0."""
    kwargs = dict(kwargs)
    if "betas" in kwargs:
        kwargs["betas"] = tuple(float(b) for b in kwargs["betas"])
    return kwargs


class Trainer:
    """State of the loop in train_celeba.py:60-77 plus the data-parallel reducers."""

    def __init__(self, cfg: TrainConfig, device: Optional[torch.device] = None, world: int = 1, z_dim: int = 128):
        self.cfg = cfg
        self.device = torch.device(device if device is not None else cfg.device)
        self.world = world
        self.z_dim = z_dim
        self.disc = build_discriminator(cfg.model, img_size=cfg.img_size, **cfg.disc_kwargs).to(self.device)
        if cfg.diff_augment:  # train_celeba.py:63 (parameter names gain the "1." prefix there too)
            from .diff_augment import DiffAugment

            self.disc = nn.Sequential(DiffAugment(), self.disc)
        self.gen = build_generator(cfg.model, img_size=cfg.img_size, **cfg.gen_kwargs).to(self.device)
        prepare_model(self.disc, spectral_norm=cfg.sn_disc, channels_last=cfg.channels_last, compile=cfg.compile)
        prepare_model(self.gen, spectral_norm=cfg.sn_gen, channels_last=cfg.channels_last, compile=cfg.compile)
        if world > 1:
            self._broadcast_parameters()
        # gradient arenas + bucketed all-reduce (single process: the hooks are inert, arenas still used)
        self.red_d = GradBucketReducer(self.disc)
        self.red_g = GradBucketReducer(self.gen)
        self.gen.bulk_noise = bool(cfg.bulk_noise)  # (read by StyleGan2Generator.forward; other generators ignore it)
        self.gen_ema = EMA(self.gen)
        self.criterion = get_loss(cfg.method)
        self.regularizer = get_regularizer(cfg.regularizer)
        okw = _fix_betas(cfg.optimizer_kwargs)
        self._graph = None
        self._graph_reals: List[Tensor] = []
        self._graph_logs: Optional[Dict[str, Tensor]] = None
        self._defer_g = bool(cfg.hip_graph and (world > 1 or cfg.defer_g_step))
        self._pending_g = False  # the generator's averaged gradients wait for their optimizer step (see flush)
        self._inflight = None
        self._inflight_g = None
        # train_celeba.py:75-77: any torch.optim class by name; Adam / AdamW run as one kernel over the flat arenas
        self.optim_d = make_optimizer(cfg.optimizer, self.disc, cfg.lr, okw, capturable=cfg.hip_graph)
        self.optim_g = make_optimizer(cfg.optimizer, self.gen, cfg.lr, okw, capturable=cfg.hip_graph)
        self.autocast_ctx = lambda: torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=cfg.mixed_precision)
        self.step = 0
        self._d_params = [p for p in self.disc.parameters() if p.requires_grad]
        self._conv_w_g = [p for p in self.gen.parameters() if p.dim() == 4]
        self._conv_w_d = [p for p in self.disc.parameters() if p.dim() == 4]

    # ---- checkpoint / resume (absent in the reference) ------------------------------------------------
    def state_dict(self) -> Dict:
        """Everything a resumed run needs: both networks, the EMA copy with its step counter (utils.py:80), both optimizers,
        the iteration count.  Tensors and plain numbers only: loads with torch.load(weights_only=True)."""
        self.flush()
        return {"step": self.step, "gen": self.gen.state_dict(), "disc": self.disc.state_dict(), "gen_ema": self.gen_ema.state_dict(),
                "optim_g": self.optim_g.state_dict(), "optim_d": self.optim_d.state_dict()}

    def load_state_dict(self, state: Dict) -> None:
        self._pending_g, self._inflight_g = False, None  # (a pending update belongs to the state being replaced)
        self.gen.load_state_dict(state["gen"])
        self.disc.load_state_dict(state["disc"])
        self.gen_ema.load_state_dict(state["gen_ema"])
        self.optim_g.load_state_dict(state["optim_g"])
        self.optim_d.load_state_dict(state["optim_d"])
        self.step = int(state["step"])
        # the parameters were overwritten in place (their arenas stay where the captured graphs read them), but through
        # copies that packed GEMM images do not notice
        GF.drop_packed()

    def save_checkpoint(self, path: str) -> None:
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path: str) -> None:
        self.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def _broadcast_parameters(self) -> None:
        import torch.distributed as dist

        for m in (self.disc, self.gen):
            for t in list(m.parameters()) + list(m.buffers()):
                broadcast(t.data, src=0)

    def train_step(self, next_reals: Callable[[], Tensor]) -> Dict[str, Tensor]:
        """One generator iteration = the `while` body of train_celeba.py:111-149.  `next_reals()` returns
        this rank's (local_batch // grad_accum, 3, S, S) real images, already on the device.

        With cfg.hip_graph the iteration is captured once, after `graph_warmup` eager iterations, into HIP
        graphs and replayed from then on: its ~2600 kernel launches cost no host time, which is what bounds
        the step at small per-GPU batches (30 ms at batch 32 launched from Python).  The graphs are cut where
        the gradients are averaged across ranks -- those all-reduces (one flat arena each) and the EMA update,
        whose weight follows a host-side schedule (utils.py:85-96), stay ordinary launches between replays."""
        if self.cfg.hip_graph:
            logs = self._graph_step(next_reals)
        else:
            logs = self._run_eager(next_reals)
        if not self._pending_g:  # (deferred: the EMA update follows the generator's optimizer step, in _finish_g)
            self.gen_ema.step()
        self.step += 1
        return logs

    def flush(self) -> None:
        """Settle a deferred generator update (cfg.defer_g_step / multi-rank HIP-graph runs): wait for the gradient
        all-reduce left in flight by the last train_step, apply the optimizer step and the EMA update.  A no-op otherwise.
        Called by state_dict(); call it before reading `gen` / `gen_ema` directly (sampling, evaluation, end of training)."""
        self._finish_g()

    def _finish_g(self) -> None:
        if not self._pending_g:
            return
        work, self._inflight_g = self._inflight_g, None
        if work is not None:
            work.wait()  # orders the current stream behind the collective; the host does not block
        self.optim_g.step()
        self.gen_ema.step()
        self._pending_g = False

    # ---- eager: gradient buckets are all-reduced from hooks while the backward pass is still running ----
    def _run_eager(self, next_reals: Callable[[], Tensor]) -> Dict[str, Tensor]:
        self._finish_g()
        if self._graph is not None:  # after graph replays the arenas still hold the last gathered gradients
            self.red_d.zero_grad()
            self.red_g.zero_grad()
        it = self._iteration(next_reals, overlap=True)
        try:
            while True:
                next(it)
        except StopIteration as stop:
            return stop.value

    def _flat_allreduce(self, red: GradBucketReducer, async_op: bool = False):
        """Mean of the whole gradient arena across ranks (one collective: 40 MB per network).  async_op: returns the
        work handle (None where the backend completed it synchronously); the caller's stream is not made to wait."""
        if red.world > 1 or red._force:
            red.arena.grad.div_(red.world)
            return all_reduce_sum(red.arena.grad, group=red.group, async_op=async_op)
        return None

    def _exchange(self, action: str, red: GradBucketReducer) -> None:
        """The gradient exchange the iteration asks for at a cut between two graph segments (see _iteration)."""
        if action == "reduce":
            self._flat_allreduce(red)
        elif action == "start":
            self._inflight = self._flat_allreduce(red, async_op=True)
        elif action == "wait":
            work, self._inflight = self._inflight, None
            if work is not None:
                work.wait()  # orders the current stream behind the collective; the host does not block
        elif action == "start_g":
            self._inflight_g = self._flat_allreduce(red, async_op=True)
            self._pending_g = True
        elif action == "finish_g":
            self._finish_g()
        else:
            raise ValueError(f"unknown gradient exchange {action!r}")

    # ---- HIP graphs ---------------------------------------------------------------------------------
    def _graph_step(self, next_reals: Callable[[], Tensor]) -> Dict[str, Tensor]:
        cfg = self.cfg
        if self._graph is None:
            if self.step < cfg.graph_warmup:
                return self._run_eager(next_reals)
            n_fetch = cfg.n_disc * cfg.grad_accum
            self._graph_reals = [next_reals().clone() for _ in range(n_fetch)]
            GF.grow_zero_pool()
            GF.prepack_weights(self._conv_w_g)  # builds the groups' tables (a host-to-device copy: not capturable)
            GF.prepack_weights(self._conv_w_d)
            GF.drop_packed()  # every packed weight the graphs read must be packed inside them
            feed = iter(self._graph_reals)
            # `.detach()`: an alias, so that requires_grad_ stays off the static buffer
            it = self._iteration(lambda: next(feed).detach(), overlap=False, defer_g=self._defer_g)
            segments = []  # (graph, (action, reducer) of the gradient exchange that follows it | None)
            pool = torch.cuda.graph_pool_handle()
            torch.cuda.synchronize()
            # With a process group up, RCCL's watchdog thread polls the events of outstanding collectives.  Under
            # the default "global" capture mode such a hipEventQuery from another thread is an error that takes the
            # process down (seen: "operation not permitted when stream is capturing" from ProcessGroupNCCL's
            # watchdog).  Capture in thread-local mode (rehearsed with and without a pause in front: not needed).
            import torch.distributed as dist

            done = False
            failed: Optional[BaseException] = None
            try:
                while not done:
                    graph = torch.cuda.CUDAGraph()
                    cut = None
                    with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                        try:
                            cut = next(it)
                        except StopIteration as stop:
                            self._graph_logs = stop.value
                            done = True
                    segments.append((graph, cut))
                    if cut is not None and cut[0] == "start_g":  # the deferred iteration ends on this cut: collect its logs
                        try:
                            next(it)
                        except StopIteration as stop:
                            self._graph_logs = stop.value
                        done = True
            except Exception as exc:  # capture refused (driver / allocator state): keep training, launched from the host
                failed = exc
                it.close()
            if dist.is_available() and dist.is_initialized():
                # all ranks replay graphs or none does: the two paths issue different collectives (one flat
                # all-reduce per network against one per bucket), so a lone fallback would hang the job
                ok = torch.tensor([0.0 if failed is not None else 1.0], device=self.device if dist.get_backend() != "gloo" else "cpu")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) == 0.0 and failed is None:
                    failed = RuntimeError("another rank could not capture the iteration")
            if failed is not None:
                import sys

                print(f"gan_amd: HIP graph capture failed ({type(failed).__name__}: {failed}); continuing without graphs", file=sys.stderr)
                self.cfg.hip_graph = False
                self._defer_g = False
                self.disc.requires_grad_(True)
                # what the aborted capture left behind was recorded, never executed: packed weights and padded inputs
                # that point at graph-pool buffers, gradient tensors handed over by a backward pass that did not run
                GF.drop_packed()
                for red in (self.red_d, self.red_g):
                    red.arena.attach_grads()
                    red.zero_grad()
                torch.cuda.synchronize()
                feed = iter(self._graph_reals)
                return self._run_eager(lambda: next(feed).detach())
            self._graph = segments
            # capture does not execute: the replay below runs the iteration on the data just fetched
        else:
            for buf in self._graph_reals:
                buf.copy_(next_reals())
        for graph, cut in self._graph:
            graph.replay()
            if cut is not None:
                self._exchange(*cut)
        GF.drop_packed()  # a replay updates the weights without bumping their Python-side versions
        return {k: v.clone() for k, v in self._graph_logs.items()}

    def _iteration(self, next_reals: Callable[[], Tensor], overlap: bool, defer_g: bool = False):
        """The iteration as a generator.  overlap=True (host-launched): gradient buckets are all-reduced from hooks
        while the backward pass runs, nothing is yielded.  overlap=False (HIP graphs): it yields (action, reducer) at
        the points where gradients cross ranks -- the cuts between the graphs:
            ("start", red_d)    after the last backward of a discriminator step: its arena's all-reduce is launched
                                asynchronously
            ("wait", red_d)     after the generator forward the NEXT pass needs -- the next discriminator step's fakes,
                                or the G step's own forward -- which does not depend on D and so runs under that
                                collective (same random draws in the same order: only stream order moves)
            ("reduce", red_g)   the generator's exchange as a plain blocking all-reduce (defer_g = False), or
            ("start_g", red_g)  launched asynchronously at the very end of the iteration, the optimizer step and the EMA
                                update left pending (defer_g = True), until
            ("finish_g", red_g) of the next iteration: placed after its first discriminator forward on the reals, the
                                one piece of the coming iteration that does not read the generator.
        Why not buckets under the backward here too: with R1 every D parameter's gradient has a first-order part
        (produced head -> input) and a second-order part (produced input -> head), so no bucket is complete until
        the backward is nearly over; and G's large tensors (512-channel blocks) are the LAST its backward reaches.
        Returns the log dict."""
        cfg, disc, gen = self.cfg, self.disc, self.gen

        def clip_d():  # train_celeba.py:115-118
            if cfg.method == "wgan" and cfg.regularizer == "none":
                with torch.no_grad():
                    for p in disc.parameters():
                        p.clip_(-0.01, 0.01)

        # every GEMM image of both networks' conv weights in one launch each (the generator's changed at the end of
        # the last iteration; the discriminator's are current unless the caches were dropped, as before a capture --
        # so a replayed iteration starts by packing what it reads, whoever changed the weights)
        GF.prepack_weights(self._conv_w_d)
        hoisted = None  # (reals, D(reals)) of the first discriminator pass, run ahead of the generator's pending update
        if defer_g and not overlap:
            clip_d()
            reals = next_reals()
            reals.requires_grad_()
            with self.autocast_ctx():
                hoisted = (reals, disc(reals))
            yield ("finish_g", self.red_g)
        GF.prepack_weights(self._conv_w_g)
        early = None    # (zs, fakes) of the coming discriminator step, computed under the previous step's all-reduce
        early_g = None  # generator output of the G step's first pass, computed under the last D step's all-reduce
        d_losses = []   # loss_d of every discriminator pass of the iteration (n_disc x grad_accum values)
        for i in range(cfg.n_disc):
            cached_reals: Optional[List[Tensor]] = [] if i == cfg.n_disc - 1 else None
            if hoisted is None:
                clip_d()
            for a in range(cfg.grad_accum):
                d_reals = None
                if hoisted is not None:
                    (reals, d_reals), hoisted = hoisted, None
                else:
                    reals = next_reals()
                    reals.requires_grad_()
                if cached_reals is not None:
                    cached_reals.append(reals.detach())  # an alias (only the relativistic loss reads it back)
                if early is not None:
                    (zs, fakes), early = early, None
                else:
                    zs = torch.randn(reals.shape[0], self.z_dim, device=self.device)
                    with self.autocast_ctx(), torch.no_grad():
                        fakes = gen(zs)
                with self.autocast_ctx():
                    if d_reals is None:
                        d_reals = disc(reals)
                    d_fakes = disc(fakes)
                    loss_d = self.criterion.d_loss(d_reals, d_fakes)
                    if self.regularizer is not None:
                        loss_d = loss_d + self.regularizer(disc, reals, fakes, d_reals)
                last = a == cfg.grad_accum - 1
                if last and overlap:
                    self.red_d.prepare()
                if a == 0 and not overlap:
                    self.red_d.arena.begin_direct()
                d_losses.append(loss_d.detach())
                with GF.direct_param_grads(not overlap):  # (the hook-driven exchange needs autograd's accumulation step)
                    # inputs = the parameters: `reals` requires grad for the regularizer (train_celeba.py:121), and a plain
                    # backward() would also carry both gradient paths all the way into the image (reals.grad, which nobody reads)
                    loss_d.backward(inputs=self._d_params)
                if last:
                    if overlap:
                        self.red_d.finish()
                    else:
                        self.red_d.arena.gather_grads()
                        yield ("start", self.red_d)
                        zs = torch.randn(reals.shape[0], self.z_dim, device=self.device)
                        with self.autocast_ctx():
                            if i == cfg.n_disc - 1:
                                early_g = gen(zs)
                            else:
                                with torch.no_grad():
                                    early = (zs, gen(zs))
                        yield ("wait", self.red_d)
            self.optim_d.step()
            GF.prepack_weights(self._conv_w_d)
            if overlap:
                self.red_d.zero_grad()

        disc.requires_grad_(False)
        try:
            for a in range(cfg.grad_accum):
                if early_g is None:
                    zs = torch.randn(reals.shape[0], self.z_dim, device=self.device)
                with self.autocast_ctx():
                    if early_g is None:
                        g_out = gen(zs)
                    else:
                        g_out, early_g = early_g, None
                    loss_g = self.criterion.g_loss(disc(g_out), lambda: disc(cached_reals[a]))
                last = a == cfg.grad_accum - 1
                if last and overlap:
                    self.red_g.prepare()
                if a == 0 and not overlap:
                    self.red_g.arena.begin_direct()
                with GF.direct_param_grads(not overlap), GF.zeroed_scratch(self.device):
                    loss_g.backward()
                if last and overlap:
                    self.red_g.finish()
        finally:
            disc.requires_grad_(True)
        # tensors, not floats: no host sync here (the reference syncs every 50 steps, train_celeba.py:153-160)
        logs = {"loss/d": loss_d.detach(), "d/real": d_reals.detach().mean(), "d/fake": d_fakes.detach().mean(), "loss/g": loss_g.detach(),
                "loss/d_steps": torch.stack([t.float().reshape(()) for t in d_losses])}
        if not overlap:
            self.red_g.arena.gather_grads()
            if defer_g:
                yield ("start_g", self.red_g)
                return logs
            yield ("reduce", self.red_g)
        self.optim_g.step()
        if overlap:
            self.red_g.zero_grad()
        return logs


def synthetic_reals(cfg: TrainConfig, device: torch.device, local_batch: int, seed: int = 0, rank: int = 0, world: int = 1) -> Callable[[], Tensor]:
    """CelebA-shaped stand-in: N(0,1) images (the real pipeline normalises to about [-1, 1],
    train_celeba.py:79-87), a small device-resident pool cycled so no host work sits in the step.
    With world > 1 every rank draws the SAME global batches (one seed) and keeps its `interleaved_shard` of each:
    MinibatchStdDev then sees the global batch's groups (progressive_gan.py:27) and N ranks train on exactly the
    data one rank would."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rows = interleaved_shard(local_batch * world, rank, world) if world > 1 else None
    pool = []
    for _ in range(4):
        batch = torch.randn(local_batch * world, 3, cfg.img_size, cfg.img_size, generator=g)
        pool.append((batch if rows is None else batch[rows]).to(device))
    if cfg.channels_last:
        pool = [p.contiguous(memory_format=torch.channels_last) for p in pool]
    state = {"i": 0}

    def next_reals() -> Tensor:
        state["i"] += 1
        return pool[state["i"] % len(pool)].clone()

    return next_reals


def dump_samples(trainer: "Trainer", fixed_zs: Tensor, log_img_dir: str, step: int) -> List[str]:
    """train_celeba.py:162-168: the fixed latents through `gen` and `gen_ema`, as 10 x 10 PNG grids.  Runs outside the captured
    graphs; a deferred generator update is settled first."""
    from .imageio import sample_grid, write_png

    trainer.flush()
    paths = []
    for suffix, model in (("", trainer.gen), ("_ema", trainer.gen_ema)):
        with trainer.autocast_ctx(), torch.no_grad():
            fakes = model(fixed_zs)
        path = os.path.join(log_img_dir, f"step{step // 1000:04d}k{suffix}.png")
        write_png(sample_grid(fakes), path)
        paths.append(path)
    GF.drop_packed()  # (packed images of the EMA weights must not outlive the next EMA update)
    return paths


def run_fid(trainer: "Trainer", fid_scorer, ref_stats, batch_size: int) -> float:
    """train_celeba.py:170-179: FID of gen_ema samples against precomputed statistics (gan_amd/fid.py)."""
    trainer.flush()

    def closure():
        zs = torch.randn(batch_size, trainer.z_dim, device=trainer.device)
        with trainer.autocast_ctx(), torch.no_grad():
            return trainer.gen_ema(zs).float()

    stats = fid_scorer.compute_stats(closure)
    GF.drop_packed()
    return float(fid_scorer.fid_score(ref_stats, stats))


def main(argv=None, fid=None) -> None:
    """fid: optional (fid_scorer, reference_stats) pair (gan_amd.fid.FID with a feature network, and the data set's statistics)."""
    args = make_parser(TrainConfig).parse_args(argv)
    cfg = TrainConfig(**vars(args))
    rank, world, device = init_from_env(cfg.device)
    if rank == 0:
        print("Config:")
        for k, v in vars(cfg).items():
            print(f"  {k}: {v}")
    if cfg.batch_size % (world * cfg.grad_accum) != 0:
        raise ValueError("batch_size must divide evenly over ranks and grad_accum")
    local_batch = cfg.batch_size // world // cfg.grad_accum
    torch.manual_seed(1234 + rank)
    trainer = Trainer(cfg, device, world)
    if cfg.resume:
        trainer.load_checkpoint(cfg.resume)
    fixed_zs = torch.randn(100, trainer.z_dim, device=device, generator=torch.Generator(device=device).manual_seed(2024))
    log_img_dir = os.path.join("images_celeba", cfg.run_name)
    if rank == 0:
        os.makedirs(log_img_dir, exist_ok=True)
    next_reals = synthetic_reals(cfg, device, local_batch, seed=0, rank=rank, world=world)
    t0 = time.perf_counter()
    while trainer.step < cfg.n_iters:
        logs = trainer.train_step(next_reals)
        step = trainer.step
        if step % 50 == 0 and rank == 0:
            if device.type == "cuda":
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rec = {k: float(v) if v.numel() == 1 else v.tolist() for k, v in logs.items()}
            rec.update(step=step, img_per_s=50 * cfg.batch_size / dt)
            print(json.dumps(rec), flush=True)
            t0 = time.perf_counter()
        if cfg.log_img_interval > 0 and step % cfg.log_img_interval == 0:
            trainer.flush()  # (every rank: the flush ends a collective)
            if rank == 0:
                dump_samples(trainer, fixed_zs, log_img_dir, step)
        if fid is not None and cfg.fid_interval > 0 and step % cfg.fid_interval == 0:
            score = run_fid(trainer, fid[0], fid[1], cfg.batch_size)
            if rank == 0:
                print(json.dumps({"step": step, "fid/ema": score}), flush=True)
        if cfg.checkpoint_interval > 0 and step % cfg.checkpoint_interval == 0:
            trainer.flush()
            if rank == 0:
                trainer.save_checkpoint(os.path.join(log_img_dir, f"step{step:07d}.pt"))
    trainer.flush()


if __name__ == "__main__":
    main()
