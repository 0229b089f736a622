"""Micro-driver for one library op (profiling aid): python tools/bench_op.py wgrad|fwd|fwd_up2|fwd_pool N Cin H W Cout k [iters]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan_amd.functional as GF
op = sys.argv[1]
n, cin, h, w, cout, k = map(int, sys.argv[2:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = torch.device('cuda:0')
# GAN_BENCH_ROTATE=R: R sets of operands used in turn, so that a launch finds its inputs in HBM as it does inside the train
# step (one set stays in the 256 MB infinity cache between launches and flatters every kernel)
rot = int(os.environ.get("GAN_BENCH_ROTATE", "1"))
xs = [torch.randn(n, cin, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(rot)]
gys = [torch.randn(n, cout, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(rot)]
auxs = [torch.randn(n, cout, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(rot)]
wgt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
pad = (k - 1) // 2
turn = [0]
def run():
    turn[0] += 1
    x, gy = xs[turn[0] % rot], gys[turn[0] % rot]
    if op == 'wgrad':
        return GF.raw_conv2d_wgrad(x, gy, k, k, pad, pad)
    if op == 'fwd':
        return GF.raw_conv2d(x, wgt, pad, pad)
    if op == 'fwd_ar':  # the input-gradient stages: mask by an activation tensor, add a residual
        return GF.raw_conv2d(x, wgt, pad, pad, aux=gy, aux_slope=0.2, resid=auxs[turn[0] % rot])
    if op == 'fwd_a':
        return GF.raw_conv2d(x, wgt, pad, pad, aux=gy, aux_slope=0.2)
    if op == 'fwd_up2':
        return GF.raw_conv2d(x, wgt, pad, pad, up2=True)
    if op == 'fwd_pool':
        return GF.raw_conv2d(x, wgt, pad, pad, pool=True)
    raise SystemExit(op)
for _ in range(3): run()
torch.cuda.synchronize()
# GAN_BENCH_FLUSH=MB: every launch timed on its own, after a fill of that many MB has swept L2 and the infinity cache
flush_mb = int(os.environ.get("GAN_BENCH_FLUSH", "0"))
if flush_mb:
    junk = torch.zeros(flush_mb << 18, device=dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a, b) in enumerate(evs):
        if os.environ.get("GAN_BENCH_FLUSH_MODE", "write") == "read":
            sink = junk.sum()  # leaves the caches full of CLEAN lines: no write-backs under the timed launch
        else:
            junk.fill_(float(i))
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    ms = ts[len(ts) // 2]
    print(f"flushed {flush_mb} MB before each launch: min {ts[0]*1e3:.1f} median {ms*1e3:.1f} max {ts[-1]*1e3:.1f} us")
else:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
hw = 4 * h * w if op == 'fwd_up2' else h * w
print(f"{op} {n}x{cin}x{h}x{w}->{cout} k{k}: {ms*1e3:.1f} us  {2.0*n*hw*cout*cin*k*k/ms/1e9:.1f} TFLOP/s")
