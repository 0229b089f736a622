"""Python call sites of a raw_* wrapper in one train iteration (profiling aid): python tools/who_calls.py raw_chan_dot [batch]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1]
batch = sys.argv[2] if len(sys.argv) > 2 else "8"
sys.argv = ["bench.py", "--no_graph", "--global_batch", batch]
import torch
import bench
import gan_amd.functional as GF
from gan_amd.train_celeba import synthetic_reals

args = bench.parse()
dev = torch.device("cuda:0")
cfg, tr, micro = bench.make_trainer(args, dev, 1)
nxt = synthetic_reals(cfg, dev, micro, seed=0)
tr.train_step(nxt)
sites = collections.Counter()
orig = getattr(GF, name)
def spy(*a, **k):
    st = [f for f in traceback.extract_stack()[:-1] if "gan_amd" in f.filename][-6:]
    shape = tuple(a[0].shape) if hasattr(a[0], "shape") else None
    sites[(shape, ("create_graph " if torch.is_grad_enabled() else "") + " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(st)))] += 1
    return orig(*a, **k)
setattr(GF, name, spy)
tr.train_step(nxt)
torch.cuda.synchronize()
for (shape, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(n, shape, site)
