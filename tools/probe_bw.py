import torch
dev='cuda:0'
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for mb in (512, 1024, 2048):
    x=torch.empty(mb*1024*1024//2, dtype=torch.bfloat16, device=dev)
    y=torch.empty_like(x)
    q=torch.empty(x.numel()//4, dtype=torch.bfloat16, device=dev)
    ms=t(lambda: x.zero_()); print(f"fill {mb} MB: {mb/1024/ms*1e3*1.0737:.0f} GB/s")
    ms=t(lambda: y.copy_(x)); print(f"copy {mb} MB: {2*mb/1024/ms*1e3*1.0737:.0f} GB/s (r+w)")
    ms=t(lambda: torch.sum(x.view(torch.int16))); print(f"read-reduce {mb} MB: {mb/1024/ms*1e3*1.0737:.0f} GB/s")
