"""Debug aid: per-tensor gradient errors of the 256x256 generator against the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import gan_oracle as O
from oracle.detparams import det_state, det_uniform
from oracle.tape import RandTape
from gan_amd.modelling import build_generator

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

dev = torch.device("cuda:0")
state = det_state(O.stylegan2_generator_shapes(256, base_dim=16), seed=22, bias_amp=0.1, scale_amp=0.05)
z = det_uniform((4, 128), 321) * 3.0
wgt = det_uniform((4, 3, 256, 256), 322) + 0.75
for mixed in (False, True):
    gen = build_generator("stylegan2", img_size=256, base_dim=16); gen.load_state_dict(state); gen = gen.to(dev)
    with RandTape() as rec:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mixed):
            img = gen(z.to(dev))
    ((img.float() * wgt.to(dev)).sum() / 1000.0 + img.float().square().mean()).backward()
    got = {k: p.grad.double().cpu() for k, p in gen.named_parameters()}
    def oracle(dtype):
        sd = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in state.items()}
        io = O.generator(sd, z.to(dtype), [t.to(dtype) for t in rec.tape])
        ((io * wgt.to(dtype)).sum() / 1000.0 + io.square().mean()).backward()
        return {k: v.grad.double() for k, v in sd.items()}
    ref = oracle(torch.float64); c32 = oracle(torch.float32)
    den = sum(float(r.square().sum()) for r in ref.values())
    rows = []
    for k in ref:
        rows.append((float((got[k] - ref[k]).square().sum()) / den, k, float(ref[k].norm()), rel(got[k], ref[k]), rel(c32[k], ref[k])))
    rows.sort(reverse=True)
    print(f"mixed={mixed}: total {sum(r[0] for r in rows) ** 0.5:.2e}")
    for share, k, nrm, e, ec in rows[:14]:
        print(f"   {k:36s} share {share ** 0.5:.2e} |ref| {nrm:.3e} hip {e:.2e} cpu32 {ec:.2e}")
    rows.sort(key=lambda r: -r[3])
    print("  worst rel:")
    for share, k, nrm, e, ec in rows[:10]:
        print(f"   {k:36s} share {share ** 0.5:.2e} |ref| {nrm:.3e} hip {e:.2e} cpu32 {ec:.2e}")
