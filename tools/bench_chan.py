"""Micro-driver for the channel reductions (profiling aid): python tools/bench_chan.py N C H W [b_mode] [iters]"""
import sys, torch
sys.path.insert(0, '.')
import gan_amd.functional as GF
n, c, h, w = map(int, sys.argv[1:5])
b_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 30
dev = torch.device('cuda:0')
A = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
B = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) if b_mode == 1 else None
run = lambda: GF.raw_chan_dot(A, B, b_mode, None, 1.0, True)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"chan_dot {n}x{c}x{h}x{w} mode {b_mode}: {ms*1e3:.1f} us  {A.numel()*2*(2 if b_mode==1 else 1)/ms/1e6:.1f} GB/s")
