"""Count the aten ops / device kernels of one training iteration, with the Python line that issues each
(profiling aid): python tools/op_count.py"""
import sys, collections, torch
sys.path.insert(0, '.')
sys.argv = ['bench.py', '--no_graph']
import bench
from gan_amd.train_celeba import synthetic_reals
from torch.profiler import profile, ProfilerActivity

args = bench.parse()
dev = torch.device('cuda:0')
cfg, tr, micro = bench.make_trainer(args, dev, 1)
nxt = synthetic_reals(cfg, dev, micro, seed=0)
for _ in range(2):
    tr.train_step(nxt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    tr.train_step(nxt)
    torch.cuda.synchronize()
by_src = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    kt = sum(k.duration for k in ev.kernels)
    if not ev.kernels:
        continue
    key = (ev.name, str(ev.input_shapes)[:90])
    by_src[key] += len(ev.kernels)
    dur[key] += kt
rows = sorted(by_src, key=lambda k: -dur[k])[:90]
for k in rows:
    print(f"{by_src[k]:5d} {dur[k]/1e3:8.3f} ms  {k[0]:28s} {k[1]}")
print('total kernels from aten ops', sum(by_src.values()), 'ms', sum(dur.values()) / 1e3)
