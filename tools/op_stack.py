"""Python call sites of the small aten ops in one iteration (profiling aid): python tools/op_stack.py [op ...]"""
import sys, collections, torch
sys.path.insert(0, '.')
ops = set(sys.argv[1:]) or {"aten::copy_", "aten::add_", "aten::add", "aten::sum", "aten::fill_", "aten::zero_", "aten::mul"}
sys.argv = ['bench.py', '--no_graph', '--global_batch', '32']
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
    for cut in tr._iteration(nxt, overlap=False):  # the op sequence the HIP graphs capture
        tr._exchange(*cut)
    torch.cuda.synchronize()
agg = collections.Counter()
tot = collections.Counter()
for ev in prof.key_averages(group_by_input_shape=True):
    if not ev.key.startswith('aten::') or ev.device_time_total <= 0:
        continue
    tot[ev.key] += ev.count
    if ev.key in ops:
        agg[(ev.key, str(ev.input_shapes)[:100])] += ev.count
print(tot.most_common(40))
for k, v in agg.most_common(70):
    print(f"{v:4d} {k[0]:14s} {k[1]}")
