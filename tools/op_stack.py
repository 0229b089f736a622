"""Python call sites of the aten ops on image-sized tensors in one iteration (profiling aid): python tools/op_stack.py"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(nxt)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::') or not ev.kernels:
        continue
    kt = sum(k.duration for k in ev.kernels)
    if kt < 20:
        continue
    st = [s for s in (ev.stack or []) if 'gan_amd' in s or 'bench.py' in s][:3]
    key = (ev.name, str(ev.input_shapes)[:60], ' <- '.join(s.split('/')[-1][:60] for s in st))
    agg[key][0] += 1
    agg[key][1] += kt
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:3d} {v[1]/1e3:7.3f} ms {k[0]:16s} {k[1]:60s} {k[2]}")
