"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

fp32 rounding envelope of the golden 10-step trajectory (tests/golden/trajectory16.pt).

The hinge + R1 + Adam(beta1 = 0) iteration at 16x16 / batch 4 is chaotic: the SAME oracle code run in
fp32 and in fp64, from the same parameters, reals and recorded random draws, drifts apart by 1e-4
(relative) within 3 steps and by O(1) within 10-20 steps (README lr 2e-4; Adam normalises every
gradient element to a step of ~3*lr whatever its size, and hinge terms switch on and off).  A free-running
trajectory can therefore only be compared step by step inside that envelope.  This script records the
envelope |oracle fp32 - oracle fp64| per step; tests/test_gpu_models.py::test_trajectory16_golden
requires the HIP path to stay within 1e-4 relative OR 10x the envelope of the reference's own
losses.  It needs neither the reference nor a GPU:

    python oracle/make_envelope.py        # writes tests/golden/trajectory16_envelope.pt
"""
from __future__ import annotations

import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from oracle import gan_oracle as O  # noqa: E402
from oracle.detparams import det_state, det_uniform  # noqa: E402
from oracle.tape import RandTape  # noqa: E402


def run(dtype, g):
    g_sd = det_state(O.stylegan2_generator_shapes(16, base_dim=128), seed=g["gen_seed"], bias_amp=0.0, scale_amp=0.05)
    d_sd = det_state(O.stylegan2_discriminator_shapes(16, base_dim=16), seed=g["disc_seed"], bias_amp=0.0, scale_amp=0.5)
    g_sd = {k: v.to(dtype).requires_grad_(True) for k, v in g_sd.items()}
    d_sd = {k: v.to(dtype).requires_grad_(True) for k, v in d_sd.items()}
    opt_d = torch.optim.Adam(list(d_sd.values()), 2e-4, betas=(0.0, 0.9))
    opt_g = torch.optim.Adam(list(g_sd.values()), 2e-4, betas=(0.0, 0.9))
    out = []
    for step in range(10):
        reals = (det_uniform((4, 3, 16, 16), g["reals_salt0"] + step) * 2.0).to(dtype)
        with RandTape([t.to(dtype) for t in g["tape"][step]]):
            logs = O.train_iteration(g_sd, d_sd, opt_g, opt_d, lambda: reals.clone(), method="hinge", reg="r1")
        out.append([logs["loss_d"], logs["loss_g"]])
    return torch.tensor(out, dtype=torch.float64)


def main():
    g = torch.load(os.path.join(REPO, "tests", "golden", "trajectory16.pt"), weights_only=True)
    a, b = run(torch.float32, g), run(torch.float64, g)
    env = (a - b).abs()
    for s in range(10):
        print(f"step {s}: fp32 {a[s].tolist()}  fp64 {b[s].tolist()}  |diff| {env[s].tolist()}")
    torch.save({"fp32": a, "fp64": b, "envelope": env}, os.path.join(REPO, "tests", "golden", "trajectory16_envelope.pt"))


if __name__ == "__main__":
    main()
