"""Debug aid: where does the bf16 error of the 256x256 discriminator's gradients come from?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import gan_oracle as O
from oracle.detparams import det_state, det_uniform
from gan_amd.losses import get_loss, get_regularizer
from gan_amd.modelling import build_discriminator

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

dev = torch.device("cuda:0")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
state = det_state(O.stylegan2_discriminator_shapes(size, base_dim=16), seed=21, bias_amp=0.1, scale_amp=0.5)
reals = det_uniform((4, 3, size, size), 311) * 2.0
fakes = det_uniform((4, 3, size, size), 312) * 2.0

WA = det_uniform((4,), 5) + 1.0
WB = det_uniform((4,), 6) + 1.0

def oracle(dtype, which, autocast=False):
    sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in state.items()}
    r = reals.detach().clone().to(dtype).requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        d_r = O.discriminator(sd, r)
        d_f = O.discriminator(sd, fakes.to(dtype))
        loss = 0
        if "h" in which: loss = loss + O.d_loss("hinge", d_r.float(), d_f.float())
        if "w" in which: loss = loss + O.d_loss("wgan", d_r.float(), d_f.float())
        if "p" in which: loss = loss + (d_r.float() * WA.to(d_r.device)).mean() + (d_f.float() * WB.to(d_r.device)).mean()
        if "r" in which: loss = loss + O.r1_penalty(None, r, None, d_r.float())
    loss.backward()
    return {k: (v.grad.double() if v.grad is not None else torch.zeros_like(v).double()) for k, v in sd.items()}, float(loss)

def hip(mixed, which):
    disc = build_discriminator("stylegan2", img_size=size, base_dim=16)
    disc.load_state_dict(state); disc = disc.to(dev)
    rg = reals.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
        d_r = disc(rg); d_f = disc(fakes.to(dev))
        loss = 0
        if "h" in which: loss = loss + get_loss("hinge").d_loss(d_r, d_f)
        if "w" in which: loss = loss + get_loss("wgan").d_loss(d_r, d_f)
        if "p" in which: loss = loss + (d_r.float() * WA.to(d_r.device)).mean() + (d_f.float() * WB.to(d_r.device)).mean()
        if "r" in which: loss = loss + get_regularizer("r1")(disc, rg, fakes.to(dev), d_r)
    loss.backward()
    return {k: (p.grad.double().cpu() if p.grad is not None else torch.zeros_like(p).double().cpu()) for k, p in disc.named_parameters()}, float(loss)

def total(a, b):
    num = sum(float((a[k] - b[k]).square().sum()) for k in b); den = sum(float(b[k].square().sum()) for k in b)
    return (num / den) ** 0.5

for which in ("p", "r", "pr"):
    ref, lref = oracle(torch.float64, which)
    g32, l32 = hip(False, which)
    gbf, lbf = hip(True, which)
    print(f"[{which}] loss fp64 {lref:.6f} hip32 {l32:.6f} hipbf {lbf:.6f}; total grad err hip32 {total(g32, ref):.2e} hipbf {total(gbf, ref):.2e}", flush=True)
    t0 = time.time()
    try:
        gcb, lcb = oracle(torch.float32, which, autocast=True)
        print(f"[{which}] CPU autocast-bf16 oracle: loss {lcb:.6f}, total grad err {total(gcb, ref):.2e}  ({time.time() - t0:.1f}s)", flush=True)
    except Exception as e:
        print("cpu autocast failed:", repr(e)[:200])
        gcb = None
    for k in ref:
        line = f"   {k:24s} |ref| {float(ref[k].norm()):.3e} hip32 {rel(g32[k], ref[k]):.2e} hipbf {rel(gbf[k], ref[k]):.2e}"
        if gcb is not None: line += f" cpu-bf16 {rel(gcb[k], ref[k]):.2e}"
        print(line)
