"""Micro-driver for upfirdn2d (profiling aid): python tools/bench_fir.py N C H W [up down] [iters]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan_amd.functional as GF
from gan_amd.modelling.nvidia_ops import Blur
n, c, h, w = map(int, sys.argv[1:5])
up, down = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1, 1)
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dev = torch.device('cuda:0')
x = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
if up == 1 and down == 1:
    k = torch.randn(3, 3, device=dev); args = (k, 1, 1, 1, 1, h, w); nb = 4 * x.numel()
elif up == 2:
    k = Blur.make_kernel(4).to(dev); args = (k, 2, 1, 2, 2, 2 * h, 2 * w); nb = 10 * x.numel()
else:
    k = Blur.make_kernel(4).to(dev); args = (k, 1, 2, 1, 1, h // 2, w // 2); nb = 2.5 * x.numel()
for _ in range(3): GF.raw_upfirdn2d(x, *args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): GF.raw_upfirdn2d(x, *args)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"upfirdn2d {n}x{c}x{h}x{w} up{up} down{down}: {ms*1e3:.1f} us  {nb/ms/1e6:.1f} GB/s")
