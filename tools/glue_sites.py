"""Profiling aid: every PyTorch (aten) op that launches device work in one iteration, counted by the innermost
gan_amd call site (TorchDispatchMode + the Python frame stack).   python tools/glue_sites.py [global_batch]"""
import sys, collections, os
sys.path.insert(0, '.')
batch = sys.argv[1] if len(sys.argv) > 1 else '32'
sys.argv = ['bench.py', '--no_graph', '--global_batch', batch]
import torch
import bench
from gan_amd.train_celeba import synthetic_reals
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = {'aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::detach', 'aten::alias', 'aten::expand', 'aten::t', 'aten::transpose',
        'aten::slice', 'aten::select', 'aten::unsqueeze', 'aten::squeeze', 'aten::as_strided', 'aten::permute', 'aten::split_with_sizes',
        'aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::is_pinned', 'aten::unbind', 'aten::_local_scalar_dense', 'aten::split'}
counts = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split('.')[0] if callable(getattr(func, 'name', None)) else str(func)
        out = func(*args, **(kwargs or {}))
        if name in SKIP:
            return out
        shape = None
        for a in list(args) + list((kwargs or {}).values()):
            if isinstance(a, torch.Tensor):
                if a.is_cuda:
                    shape = tuple(a.shape)
                break
            if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor) and a[0].is_cuda:
                shape = ('list', len(a))
                break
        if shape is None and not (isinstance(out, torch.Tensor) and out.is_cuda):
            return out
        try:
            f = sys._getframe(1)
        except ValueError:  # called straight from the autograd engine's thread
            f = None
        site = 'autograd engine / other'
        while f is not None:
            fn = f.f_code.co_filename
            if '/gan_amd/' in fn and not fn.endswith('_lib.py'):
                site = f"{fn[fn.index('gan_amd/'):]}:{f.f_lineno} {f.f_code.co_name}"
                break
            f = f.f_back
        counts[(site, name, shape)] += 1
        return out


args = bench.parse()
dev = torch.device('cuda:0')
cfg, tr, micro = bench.make_trainer(args, dev, 1)
nxt = synthetic_reals(cfg, dev, micro, seed=0)
for _ in range(2):
    tr.train_step(nxt)
torch.cuda.synchronize()
with Spy():
    for cut in tr._iteration(nxt, overlap=False):  # the op sequence the HIP graphs capture
        tr._exchange(*cut)
    torch.cuda.synchronize()
by_site = collections.Counter()
for (site, name, shape), n in counts.items():
    by_site[site] += n
print(f'batch {batch}: {sum(counts.values())} aten ops with device work')
for site, n in by_site.most_common(60):
    ops = collections.Counter()
    for (s2, name, shape), k in counts.items():
        if s2 == site:
            ops[f'{name[6:]}{shape}'] += k
    print(f'{n:5d}  {site:58s} ' + ', '.join(f'{k}x{v}' for k, v in ops.most_common(40 if site.startswith("autograd") else 4)))
