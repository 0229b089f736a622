"""Micro-driver for gan_modconv_bwd_pre (profiling aid): python tools/bench_bwd_pre.py N C H W [iters]"""
import sys, torch
sys.path.insert(0, '.')
import gan_amd.functional as GF
n, c, h, w = map(int, sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
dev = torch.device('cuda:0')
mk = lambda ch: torch.randn(n, ch, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
sets = [(mk(c), mk(c), mk(1)) for _ in range(4)]
dinv = torch.rand(n, c, device=dev) + 0.5
turn = [0]
def run():
    turn[0] += 1
    gy, y, nz = sets[turn[0] % 4]
    return GF.raw_modconv_bwd_pre(gy, y, nz, dinv, 0.2)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"modconv_bwd_pre {n}x{c}x{h}x{w}: {ms*1e3:.1f} us  {n*c*h*w*2*3/ms/1e6:.1f} GB/s (gy + y in, gz out)")
