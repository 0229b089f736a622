"""Micro-driver for the image-boundary kernels (profiling aid): python tools/bench_rgb.py [N H Cout]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan_amd.functional as GF
n, h, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 16)
dev = torch.device("cuda:0")
x32 = torch.randn(n, 3, h, h, device=dev)
xbf = torch.randn(n, 3, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
gy = torch.randn(n, cout, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, 3, 1, 1, device=dev)
b = torch.randn(cout, device=dev)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for tag, x in (("f32 nchw", x32), ("bf16 nhwc", xbf)):
    by = lambda *ts: sum(t.numel() * t.element_size() for t in ts)
    t = timeit(lambda: GF._raw_rgb_fwd(x, w, b, 1.0, torch.bfloat16)); print(f"fwd   {tag}: {t:7.1f} us {by(x, gy) / t / 1e3:7.1f} GB/s")
    t = timeit(lambda: GF._raw_rgb_wgrad(x, gy, bias=True)); print(f"wgrad {tag}: {t:7.1f} us {by(x, gy) / t / 1e3:7.1f} GB/s")
t = timeit(lambda: GF._raw_rgb_dgrad(gy, w)); print(f"dgrad bf16: {t:7.1f} us {(gy.numel() * 2 + n * 3 * h * h * 2) / t / 1e3:7.1f} GB/s")
