"""Profiling aid: one event-bracketed iteration, launches grouped by (family, shape tag).
python tools/by_shape.py [global_batch] [family substring] [extra bench.py flags, e.g. --recipe dcgan --img_size 64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
batch = sys.argv[1] if len(sys.argv) > 1 else '256'
pat = sys.argv[2] if len(sys.argv) > 2 else ''
sys.argv = ['bench.py', '--no_graph', '--global_batch', batch] + sys.argv[3:]
import torch
import bench
import gan_amd.functional as GF
from gan_amd.train_celeba import synthetic_reals
args = bench.parse()
dev = torch.device('cuda:0')
cfg, tr, micro = bench.make_trainer(args, dev, 1)
nxt = synthetic_reals(cfg, dev, micro, seed=0)
for _ in range(2):
    tr.train_step(nxt)
torch.cuda.synchronize()
prof = GF.KernelProfile()
GF.set_profile(prof)
tr.train_step(nxt)
torch.cuda.synchronize()
GF.set_profile(None)
rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]['ms'])
tot = sum(v['ms'] for _, v in rows)
print(f'total bracketed {tot:.2f} ms')
for (fam, tag), v in rows:
    if pat in fam:
        gbs = v['bytes'] / v['ms'] / 1e6 if v['ms'] else 0
        tf = v['flops'] / v['ms'] / 1e9 if v['ms'] else 0
        print(f"{fam:32s} {str(tag):48s} n={v['launches']:3d} {v['ms']:7.3f} ms  {gbs:7.0f} GB/s {tf:7.1f} TF/s")
