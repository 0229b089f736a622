"""Benchmark of the StyleGAN2 hinge + R1 training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): StyleGAN2 generator +
discriminator (base_dim 16 both, 10.2 M / 9.9 M parameters), 256x256, GLOBAL batch 256 split evenly over
the ranks (strong scaling), hinge loss + R1 every step, Adam(2e-4, betas (0, 0.9)), bf16 autocast +
channels_last, EMA; synthetic CelebA-shaped reals resident in HBM.  A "step" is one generator iteration
of train_celeba.py:111-149: one D update (with the R1 double backward) + one G update + EMA.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      -- the kernel family with the largest share of device time in one extra, event-bracketed
                   step after the timed region (HIP events on the launch stream around every C-ABI call)
  cpu_baseline  -- the oracle (CPU restatement of the reference, fp32) timed on the host cores on a
                   bounded sample of the same workload (rank 0, N=1 only)
  upfirdn2d     -- config[1] microbenchmark: achieved HBM GB/s of the upfirdn2d / bias+leaky-ReLU kernels
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
# profile family (gan_amd/functional.py) -> device kernel; the first three are priced against the MFMA peak
MFMA_FAMILIES = {"conv2d_fwd_wide_halo": "conv_wide_halo4_k<BN, UP2> (16-wide tiles) / conv_wide_halo_k<BN, 8, UP2> (8 x 8 maps)", "conv2d_fwd_wide128": "conv_wide_k<128, 2>", "conv2d_fwd_wide64": "conv_wide_k<64, 2> / conv_wide_k<64, 4>",
                 "conv2d_fwd": "conv_fwd_k<...>", "conv2d_wgrad": "conv_wgrad_tiled_k<...> / conv_wgrad_k<...>"}
HBM_FAMILIES = {"conv2d_fwd_thin": "conv_thin_k<...>", "conv2d_wgrad_thin": "conv_thin_wgrad_k<...>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--img_size", type=int, default=256)
    ap.add_argument("--global_batch", type=int, default=256)
    ap.add_argument("--micro_batch", type=int, default=256, help="largest per-rank batch per backward pass (grad_accum above it)")
    ap.add_argument("--base_dim", type=int, default=16)
    ap.add_argument("--recipe", choices=["stylegan2", "dcgan"], default="stylegan2",
                    help="stylegan2: hinge+R1 (BASELINE configs 3/4); dcgan: DCGAN WGAN-GP n_disc=5 at 64x64 (config 5, README.md:55)")
    ap.add_argument("--blur_size", type=int, default=None,
                    help="FIR low-pass of this size in the StyleGAN2 blocks (not a BASELINE config: the reference leaves them out)")
    ap.add_argument("--fp32", action="store_true", help="fp32 instead of bf16 autocast (debug)")
    ap.add_argument("--no_cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no_micro", action="store_true", help="skip the upfirdn2d microbenchmark leg")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel from the host instead of replaying the iteration as HIP graphs")
    ap.add_argument("--micro_only", action="store_true", help="run only the upfirdn2d microbenchmark leg and print it")
    ap.add_argument("--no_side", action="store_true", help="skip BASELINE configs 3 and 5 (extra keys of the JSON line)")
    ap.add_argument("--no_pmc", action="store_true", help="skip the PMC traffic passes (roofline.traffic stays null)")
    return ap.parse_args()


def make_trainer(args, device, world):
    from gan_amd.train_celeba import TrainConfig, Trainer

    local = args.global_batch // world
    # grad_accum (train_celeba.py:38,120): smallest split whose micro-batch fits --micro_batch and keeps
    # whole MinibatchStdDev groups of 4
    accum = next(a for a in range(1, local + 1) if local % a == 0 and local // a <= args.micro_batch and (local // a) % 4 == 0)
    if args.recipe == "dcgan":
        cfg = TrainConfig(
            model="dcgan", img_size=args.img_size, gen_kwargs={}, disc_kwargs={"norm": "none"}, method="wgan", regularizer="wgan-gp",
            n_disc=5, batch_size=args.global_batch, optimizer="Adam", optimizer_kwargs={"betas": [0.0, 0.9]}, lr=1e-4, device="cuda",
            mixed_precision=not args.fp32, channels_last=True, grad_accum=accum, hip_graph=not args.no_graph, graph_warmup=1,
        )
    else:
        extra = {} if getattr(args, "blur_size", None) is None else {"blur_size": args.blur_size}
        cfg = TrainConfig(
            model="stylegan2", img_size=args.img_size, gen_kwargs={"base_dim": args.base_dim, **extra}, disc_kwargs={"base_dim": args.base_dim, **extra},
            method="hinge", regularizer="r1", batch_size=args.global_batch, optimizer="Adam", optimizer_kwargs={"betas": [0.0, 0.9]},
            lr=2e-4, device="cuda", mixed_precision=not args.fp32, channels_last=True, grad_accum=accum,
            hip_graph=not args.no_graph, graph_warmup=1, bulk_noise=True,
        )
    return cfg, Trainer(cfg, device, world), local // accum


def roofline_leg(trainer, next_reals):
    """One extra step with every C-ABI launch bracketed by events; returns the roofline object of the
    dominant family and the per-family table."""
    import gan_amd.functional as GF

    prof = GF.KernelProfile()
    GF.set_profile(prof)
    try:
        GF.drop_packed()
        trainer._run_eager(next_reals)  # launched from the host even when the timed steps replay HIP graphs
        torch.cuda.synchronize()
    finally:
        GF.set_profile(None)
    if os.environ.get("GAN_AMD_BENCH_DETAIL") and os.environ.get("RANK", "0") == "0":
        rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]["ms"])
        with open(os.environ.pop("GAN_AMD_BENCH_DETAIL"), "w") as f:  # (the headline workload only: the side configs run after it)
            for (fam, tag), d in rows:
                sec = d["ms"] * 1e-3
                f.write(f"{fam:14s} {str(tag):44s} n={d['launches']:3d} ms={d['ms']:8.3f} TF={d['flops'] / sec / 1e12:8.1f} GB/s={d['bytes'] / sec / 1e9:8.1f}\n")
    table = prof.summary()
    total_ms = sum(d["ms"] for d in table.values())
    fams = {}
    for name, d in table.items():
        sec = d["ms"] * 1e-3
        mfma = bool(d["flops"]) and name in MFMA_FAMILIES
        fams[name] = dict(
            launches=d["launches"], ms=round(d["ms"], 3), share=round(d["ms"] / total_ms, 4),
            tflops=round(d["flops"] / sec / 1e12, 2) if d["flops"] else None,
            gbs=round(d["bytes"] / sec / 1e9, 1), bound="mfma" if mfma else "hbm",
            frac=round(d["flops"] / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS if mfma else d["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4),
        )
    # the thin direct convolution is ONE kernel template compiled per (Cin, Cout, k, mode): one aggregated row too, so that
    # the table says where the time is and not how finely a family is split
    thin = [v for k, v in table.items() if k.startswith("conv2d_fwd_thin<")]
    if thin:
        ms = sum(v["ms"] for v in thin)
        fl, by = sum(v["flops"] for v in thin), sum(v["bytes"] for v in thin)
        fams["conv2d_fwd_thin<*> (all instantiations)"] = dict(
            launches=sum(v["launches"] for v in thin), ms=round(ms, 3), share=round(ms / total_ms, 4), tflops=round(fl / ms / 1e9, 2),
            gbs=round(by / ms / 1e6, 1), bound="hbm", frac=round(by / ms / 1e6 / HBM_PEAK_GBS, 4))
    dom = max(table, key=lambda k: table[k]["ms"])
    d = table[dom]
    sec = d["ms"] * 1e-3
    # MFMA-bound families: the wide-layer implicit GEMMs and weight gradients; everything else (thin-layer
    # convolutions at 144..576 MACs per output element, FIR, pointwise, reductions) is priced against HBM
    if d["flops"] and dom in MFMA_FAMILIES:
        ach = d["flops"] / sec / 1e12
        roof = dict(kernel=dom, symbol=MFMA_FAMILIES[dom], bound="mfma", achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                    launches=d["launches"], avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2),
                    hbm_gbs=round(d["bytes"] / sec / 1e9, 1))
    else:
        ach = d["bytes"] / sec / 1e9
        roof = dict(kernel=dom, symbol=("conv_thin_k" + dom[len("conv2d_fwd_thin"):] if dom.startswith("conv2d_fwd_thin") else HBM_FAMILIES.get(dom, dom + "_k")), bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                    launches=d["launches"], avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2))
    if thin:
        roof["largest_family"] = dict(name="conv2d_fwd_thin<*> (all instantiations)", **{k: fams["conv2d_fwd_thin<*> (all instantiations)"][k]
                                                                                        for k in ("launches", "ms", "share", "gbs", "bound", "frac")})
    return roof, fams


def pmc_traffic_leg(roof):
    """HBM-side bytes per launch of the dominant kernel from the PMC counters (MI355X_MICROARCH.md, HBM section):
    FETCH_SIZE and WRITE_SIZE need a rocprofv3 pass each, so two child processes run that kernel on its most common
    shape of the workload (tools/bench_op.py) under `rocprofv3 --pmc`; FETCH_SIZE is in KB and is doubled for 16-byte
    per lane streaming reads on gfx950, WRITE_SIZE is in KB and exact.  Returns None when the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    shapes = {  # family -> (tools/bench_op.py arguments, kernel-name fragment, algorithmic bytes of that launch)
        "conv2d_fwd_wide_halo": (["fwd", "256", "128", "32", "32", "128", "3"], "conv_wide_halo4_k<128", 2 * (256 * 32 * 32 * 128) * 2 + 128 * 1152 * 2),
        "conv2d_fwd_wide128": (["fwd", "64", "128", "32", "32", "128", "3"], "conv_wide_k<128", 2 * (64 * 32 * 32 * 128) * 2 + 128 * 1152 * 2),
        "conv2d_fwd_wide64": (["fwd", "32", "512", "8", "8", "512", "3"], "conv_wide_k<64", 2 * (32 * 8 * 8 * 512) * 2 + 512 * 4608 * 2),
        "conv2d_wgrad": (["wgrad", "256", "128", "32", "32", "128", "3"], "conv_wgrad_tiled_k", 2 * (256 * 32 * 32 * 128) * 2 + 128 * 1152 * 4),
        "conv2d_wgrad_thin": (["wgrad", "256", "16", "256", "256", "16", "3"], "conv_thin_wgrad_k", 2 * (256 * 256 * 256 * 16) * 2),
    }
    if roof is None or roof["kernel"] not in shapes or shutil.which("rocprofv3") is None:
        return None
    op_args, frag, algo = shapes[roof["kernel"]]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="gan_amd_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_op.py")] + op_args + ["4"]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            got = []
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if frag in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            got.append(float(row["Counter_Value"]))
            if not got:
                return None
            vals[counter] = got[-1] * 1024.0  # KB -> bytes, the last (warm) launch
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return dict(bytes=round(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]), read_bytes=round(2.0 * vals["FETCH_SIZE"]), write_bytes=round(vals["WRITE_SIZE"]),
                algorithmic_bytes=algo, shape="tools/bench_op.py " + " ".join(op_args),
                note="per launch of " + frag + ">; FETCH_SIZE (KB) x 2 for 16-byte-per-lane reads on gfx950 + WRITE_SIZE (KB), one rocprofv3 --pmc pass each")


def side_recipe(device, recipe, img_size, base_dim, batch, steps=10, warmup=2):
    """BASELINE configs 3 and 5 beside the headline line: the same timing protocol on one GPU (graph replay, W warm-up,
    K timed steps) and the event-bracketed iteration's dominant family."""
    from gan_amd.train_celeba import synthetic_reals

    ns = argparse.Namespace(global_batch=batch, micro_batch=256, recipe=recipe, img_size=img_size, base_dim=base_dim, fp32=False, no_graph=False)
    cfg, trainer, micro = make_trainer(ns, device, 1)
    next_reals = synthetic_reals(cfg, device, micro, seed=0)
    for _ in range(cfg.graph_warmup + 1 + warmup):
        trainer.train_step(next_reals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(next_reals)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    roof, _ = roofline_leg(trainer, next_reals)
    graphed = bool(cfg.hip_graph and trainer._graph is not None)
    del trainer
    torch.cuda.empty_cache()
    return dict(img_per_s=round(batch / dt, 1), ms_per_step=round(dt * 1e3, 2), steps=steps, warmup=warmup, hip_graph=graphed, roofline=roof,
                workload=(f"DCGAN WGAN-GP n_disc=5, {img_size}x{img_size}, batch {batch}, Adam(1e-4, (0, 0.9)), bf16" if recipe == "dcgan" else
                          f"StyleGAN2 hinge+R1, {img_size}x{img_size}, base_dim {base_dim}, batch {batch}, Adam(2e-4, (0, 0.9)), bf16"))


def share_leg(device, args, batch, steps=10, warmup=2):
    """The per-GPU share of the fixed global batch at N = 256 / batch ranks, on this one GPU (graph replay, no all-reduce):
    what the strong-scaling curve can reach before any communication."""
    from gan_amd.train_celeba import synthetic_reals

    ns = argparse.Namespace(global_batch=batch, micro_batch=256, recipe="stylegan2", img_size=args.img_size, base_dim=args.base_dim, fp32=args.fp32,
                            no_graph=args.no_graph)
    cfg, trainer, micro = make_trainer(ns, device, 1)
    next_reals = synthetic_reals(cfg, device, micro, seed=0)
    for _ in range(cfg.graph_warmup + 1 + warmup):
        trainer.train_step(next_reals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(next_reals)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del trainer
    torch.cuda.empty_cache()
    return round(dt * 1e3, 2)


def micro_leg(device):
    """BASELINE config[1]: benchmark_upfirdn2d.py shapes, bf16 channels_last, 3x3 FIR, up = down = 1, pad 1;
    plus 2x up / 2x down with the 4-tap binomial FIR and the fused bias + leaky-ReLU.  Algorithmic bytes =
    (numel_in + numel_out) * 2 per call."""
    import gan_amd.functional as GF
    from gan_amd.modelling.nvidia_ops import Blur

    shapes = [(32, 512, 4, 4), (32, 512, 8, 8), (32, 512, 16, 16), (32, 512, 32, 32), (32, 256, 64, 64),
              (32, 128, 128, 128), (32, 64, 256, 256), (32, 32, 512, 512), (32, 16, 1024, 1024)]
    k3 = torch.randn(3, 3, device=device)
    k4 = Blur.make_kernel(4).to(device)
    out = {}

    def timeit(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters

    for shp in shapes:
        x = torch.randn(shp, device=device, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        n, c, h, w = shp
        numel = x.numel()
        iters = 100 if numel < (1 << 26) else 20
        key = "x".join(map(str, shp))
        rec = {}
        t = timeit(lambda: GF.raw_upfirdn2d(x, k3, 1, 1, 1, 1, h, w), iters)
        rec["fir3_gbs"] = round(4 * numel / t / 1e9, 1)
        bias = torch.zeros(c, device=device)
        t = timeit(lambda: GF.raw_affine_act(x, b=bias, slope=0.2), iters)
        rec["bias_lrelu_gbs"] = round(4 * numel / t / 1e9, 1)
        if numel <= (1 << 27):
            t = timeit(lambda: GF.raw_upfirdn2d(x, k4, 2, 1, 2, 2, 2 * h, 2 * w), iters)
            rec["up2_fir4_gbs"] = round(2 * 5 * numel / t / 1e9, 1)
        t = timeit(lambda: GF.raw_upfirdn2d(x, k4, 1, 2, 1, 1, h // 2, w // 2), iters)
        rec["down2_fir4_gbs"] = round(2 * 1.25 * numel / t / 1e9, 1)
        out[key] = rec
        del x
    # benchmark_upfirdn2d.py:15-29 as written: out = f(x) ** (n_backward + 1), then n_backward times
    # (out,) = autograd.grad(out.sum(), data, create_graph=True); 10 warm-up + N timed iterations, each followed by a
    # device synchronise; f = UpFIRDn2d.apply(data, kernel, 1, 1, 1, 1, 1, 1) (the Function whose backward re-enters
    # the same HIP kernel: no autograd-of-autograd over torch ops at any order).  it/s, the reference's unit.
    from gan_amd.modelling.nvidia_ops import UpFIRDn2d

    def its(data, kern, n_backward, n_iter):
        def one():
            o = UpFIRDn2d.apply(data, kern, 1, 1, 1, 1, 1, 1) ** (n_backward + 1)
            for _ in range(n_backward):
                (o,) = torch.autograd.grad(o.sum(), data, create_graph=True)

        for _ in range(10):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_iter):
            one()
            torch.cuda.synchronize()
        return n_iter / (time.perf_counter() - t0)

    for shp in shapes:
        data = torch.randn(shp, device=device, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        n_iter = 100 if data.numel() < (1 << 28) else 30
        rec = out["x".join(map(str, shp))]
        rec["its_per_s_n_backward"] = {str(nb): round(its(data, k3, nb, n_iter), 1) for nb in range(3)}
        rec["timed_iters"] = n_iter
        del data
        torch.cuda.empty_cache()
    # DiffAugment in front of the discriminator (diff_augment.py:54-60) at the headline batch: the reference's
    # eight kernels as one sum pass + one apply pass.  Algorithmic bytes: the image read once and written once.
    import gan_amd.diff_augment as DA

    da = {}
    for dt, tag in ((torch.float32, "f32_nchw"), (torch.bfloat16, "bf16_nhwc")):
        imgs = torch.randn(256, 3, 256, 256, device=device, dtype=dt)
        if dt == torch.bfloat16:
            imgs = imgs.contiguous(memory_format=torch.channels_last)
        usk = torch.rand(256, 3, device=device)
        shift = torch.tensor([7, -5, 40, 90], dtype=torch.int32, device=device)
        t = timeit(lambda: DA._raw_diffaug(imgs, usk, shift, 0), 50)
        da[tag] = {"us": round(t * 1e6, 1), "gbs": round(2 * imgs.numel() * imgs.element_size() / t / 1e9, 1)}
        da[tag]["frac"] = round(da[tag]["gbs"] / HBM_PEAK_GBS, 4)
        del imgs
    big = out["32x16x1024x1024"]
    return dict(shapes=out, diff_augment_256x3x256x256=da, peak_gbs=HBM_PEAK_GBS, fir3_frac_at_largest=round(big["fir3_gbs"] / HBM_PEAK_GBS, 4),
                bias_lrelu_frac_at_largest=round(big["bias_lrelu_gbs"] / HBM_PEAK_GBS, 4),
                down2_fir4_frac_at_largest=round(big["down2_fir4_gbs"] / HBM_PEAK_GBS, 4),
                up2_fir4_frac_at_32x64x256x256=round(out["32x64x256x256"]["up2_fir4_gbs"] / HBM_PEAK_GBS, 4),
                method="*_gbs: raw kernel launches, events over 100 (20) launches, algorithmic bytes (in + out) * 2; "
                       "its_per_s_n_backward: benchmark_upfirdn2d.py:15-29 on UpFIRDn2d.apply, bf16 channels_last")


def cpu_leg(args):
    """The oracle's train_iteration (fp32, reference init) on the host cores: same model and image size,
    batch 4, one warm-up + timed iterations bounded to about 20 s."""
    from oracle import gan_oracle as O

    torch.manual_seed(0)
    batch = 4
    g_sd, d_sd = O.new_stylegan2_state(args.img_size, args.base_dim, args.base_dim)
    opt_d = torch.optim.Adam(list(d_sd.values()), 2e-4, betas=(0.0, 0.9))
    opt_g = torch.optim.Adam(list(g_sd.values()), 2e-4, betas=(0.0, 0.9))
    reals = torch.randn(batch, 3, args.img_size, args.img_size)
    threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU
    torch.set_num_threads(threads)
    O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals.clone(), method="hinge", reg="r1")  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while iters < 10 and (iters == 0 or time.perf_counter() - t0 < 20.0):
        O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals.clone(), method="hinge", reg="r1")
        iters += 1
    dt = (time.perf_counter() - t0) / iters
    return dict(value=round(batch / dt, 3), unit="img/s", cores=threads, kind="port",
                sample=f"oracle (fp32 PyTorch restatement of the reference) hinge+R1 step, {args.img_size}x{args.img_size}, batch {batch}, {iters} timed iteration(s) after 1 warm-up")


_REAL_STDOUT = None


def _quiet_stdout():
    """stdout carries ONE JSON line: everything else that native libraries print there (RCCL's five-line version
    banner at communicator set-up, for one) is sent to stderr for the length of the run."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def _emit(line: dict) -> None:
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, data)


def main():
    _quiet_stdout()
    if os.environ.get("GAN_AMD_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["GAN_AMD_BENCH_WATCHDOG"]), exit=True)
    args = parse()
    from gan_amd.distributed import init_from_env
    from gan_amd.train_celeba import synthetic_reals

    rank, world, device = init_from_env("cuda")
    if args.micro_only:
        _emit(micro_leg(device))
        return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.global_batch % (4 * world) != 0:
        raise SystemExit("global batch must be a multiple of 4 * gpus (MinibatchStdDev groups)")
    torch.manual_seed(1234 + rank)
    cfg, trainer, micro = make_trainer(args, device, world)
    # one seed: every rank draws the same global batches and keeps its interleaved shard (MinibatchStdDev groups)
    next_reals = synthetic_reals(cfg, device, micro, seed=0, rank=rank, world=world)

    def sync():
        trainer.flush()  # (multi-rank graph runs leave the generator's update pending across iterations: K steps = K updates)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if cfg.hip_graph:  # set-up: one eager iteration, then the capture; not part of the W warm-up steps
        for _ in range(cfg.graph_warmup + 1):
            trainer.train_step(next_reals)
    for _ in range(args.warmup):
        trainer.train_step(next_reals)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(next_reals)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device if dist.get_backend() != "gloo" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    logs = trainer.train_step(next_reals)  # sanity: losses stay finite
    finite = all(bool(torch.isfinite(v).all()) for v in logs.values())
    peak_mem = torch.cuda.max_memory_allocated(device) / 2**30
    graphed = bool(cfg.hip_graph and trainer._graph is not None)

    roof = fams = micro_res = cpu = None
    if not args.no_roofline:
        # every rank runs the event-bracketed iteration (its gradient all-reduces need all of them); rank 0 reports
        r = roofline_leg(trainer, next_reals)
        if rank == 0:
            roof, fams = r
    if world > 1:
        dist.barrier()
    side = None
    if rank == 0 and world == 1:
        del trainer
        torch.cuda.empty_cache()
        if not args.no_side and args.recipe == "stylegan2" and args.img_size == 256:
            side = {"config3_stylegan2_64": side_recipe(device, "stylegan2", 64, 64, 64),
                    "config5_dcgan_wgan_gp": side_recipe(device, "dcgan", 64, 16, 64)}
            if args.global_batch == 256:
                # ms per iteration of the per-GPU share at 2 / 4 / 8 ranks, run on this one GPU without the all-reduce
                side["per_gpu_share_ms"] = {str(b): share_leg(device, args, b) for b in (128, 64, 32)}
        if not args.no_micro:
            micro_res = micro_leg(device)
        if roof is not None and not args.no_pmc:
            torch.cuda.synchronize()
            tr = pmc_traffic_leg(roof)
            if tr is not None:
                roof["traffic"] = tr["bytes"]
                roof["traffic_detail"] = tr
        if not args.no_cpu:
            cpu = cpu_leg(args)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": (f"DCGAN WGAN-GP n_disc=5 train img/s @{args.img_size}" if args.recipe == "dcgan" else
                       "StyleGAN2 G+D+R1 train img/s @256x256" if args.img_size == 256 else f"StyleGAN2 G+D+R1 train img/s @{args.img_size}"),
            "value": round(args.global_batch * args.steps / elapsed, 2),
            "unit": "img/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "fp32" if args.fp32 else "bf16",
            "data": "synthetic",
            "config": {
                "workload": (f"DCGAN WGAN-GP n_disc=5 train iteration, {args.img_size}x{args.img_size}, global batch {args.global_batch}, Adam, EMA"
                             if args.recipe == "dcgan" else
                             f"StyleGAN2 hinge+R1 train step, {args.img_size}x{args.img_size}, base_dim {args.base_dim}, global batch {args.global_batch}, Adam, EMA"
                             + ("" if args.blur_size is None else f", blur_size {args.blur_size} (NOT a BASELINE config)")),
                "global_batch": args.global_batch,
                "per_gpu_batch": args.global_batch // world,
                "grad_accum": cfg.grad_accum,
                "parallelism": f"dp{world}",
                "losses_finite": finite,
                "peak_mem_gib": round(peak_mem, 1),
                "hip_graph": graphed,
            },
            "roofline": roof,
            "kernels": fams,
            "cpu_baseline": cpu,
            "upfirdn2d": micro_res,
            "per_gpu_share_ms": side.pop("per_gpu_share_ms", None) if side else None,
            "other_configs": side,
        }
        _emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
