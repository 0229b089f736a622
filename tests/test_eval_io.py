"""SURVEY.md 8 f3 (host side, no kernels): FID statistics and score (reference fid.py), the 10 x 10 sample grid and its PNG
(train_celeba.py:162-168), checkpoint / resume of the Trainer (GPU)."""
import math
import os
import socket
import struct
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def test_fid_score_and_stats():
    """fid_score (fid.py:33-39, eigenvalue form) against the textbook matrix-square-root form in scipy; compute_stats
    against numpy's mean / covariance; the default constructor says what it is missing."""
    import scipy.linalg

    from gan_amd.fid import FID

    torch.manual_seed(3)
    feats = [torch.randn(400, 12, dtype=torch.float64) @ torch.randn(12, 12, dtype=torch.float64) + s for s in (0.0, 0.3)]
    stats = [dict(mean=f.mean(0), cov=f.T.cov()) for f in feats]
    got = float(FID.fid_score(*stats))
    m1, m2, c1, c2 = stats[0]["mean"].numpy(), stats[1]["mean"].numpy(), stats[0]["cov"].numpy(), stats[1]["cov"].numpy()
    want = float(((m1 - m2) ** 2).sum() + np.trace(c1) + np.trace(c2) - 2 * np.trace(scipy.linalg.sqrtm(c1 @ c2).real))
    assert math.isclose(got, want, rel_tol=1e-6), (got, want)
    assert abs(float(FID.fid_score(stats[0], stats[0]))) < 1e-6

    class Net(nn.Module):  # InceptionV3's calling convention: a list whose first element flattens to (N, F)
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 5, 3)

        def forward(self, x):
            return [self.conv(x).mean(dim=(2, 3), keepdim=True)]

    net = Net()
    fid = FID("cpu", feature_net=net, n_samples=70)
    g = torch.Generator().manual_seed(5)
    batches = [torch.randn(16, 3, 8, 8, generator=g) for _ in range(5)]
    it = iter(batches)
    st = fid.compute_stats(lambda: next(it))
    with torch.no_grad():
        ref = torch.cat([net(b)[0].flatten(1) for b in batches])[:70].numpy()
    assert np.allclose(st["mean"].numpy(), ref.mean(0), atol=1e-6) and np.allclose(st["cov"].numpy(), np.cov(ref.T), atol=1e-6)
    try:
        import pytorch_fid  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="feature_net"):
            FID("cpu")


class _Feat(nn.Module):
    def forward(self, x):
        return x.flatten(1)[:, :6] * 2.0


def _fid_worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gan_amd.fid import FID

    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(40, 3, 2, 2, generator=g)  # 3 batches of 16 > the 38 rows each rank owes for n_samples = 75
    it = iter(mine.split(16))
    st = FID("cpu", feature_net=_Feat(), n_samples=75).compute_stats(lambda: next(it))
    torch.save(dict(stats=st, mine=mine), os.path.join(out_dir, f"f{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_fid_feature_gather_gloo_world2(tmp_path):
    """SURVEY 8 f3 'distributed feature gather': two ranks evaluate 38 samples each, the features are all-gathered in rank
    order and cut to n_samples; both ranks return the statistics of that one feature matrix."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_fid_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"f{r}.pt") for r in range(2))
    feats = torch.cat([_Feat()(r["mine"][:38]) for r in (r0, r1)])[:75]
    for r in (r0, r1):
        assert torch.allclose(r["stats"]["mean"], feats.mean(0), atol=1e-6) and torch.allclose(r["stats"]["cov"], feats.T.cov(), atol=1e-6)


def test_sample_grid_and_png(tmp_path):
    from gan_amd.imageio import encode_png, sample_grid, write_png
    from gan_amd.utils import unnormalize_img

    torch.manual_seed(1)
    fakes = torch.rand(100, 3, 6, 4) * 2 - 1
    grid = sample_grid(fakes)
    assert grid.dtype == torch.uint8 and tuple(grid.shape) == (3, 60, 40)
    # sample r*10 + c sits at rows r*6 .., columns c*4 .. (train_celeba.py:166: unflatten -> permute -> flatten)
    assert torch.equal(grid[:, 12:18, 28:32], unnormalize_img(fakes[27]))
    path = tmp_path / "g.png"
    write_png(grid, str(path))
    data = path.read_bytes()
    assert data[:8] == bytes([0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A])
    pos, chunks = 8, {}
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks[tag] = body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (40, 60, 8, 2) and b"IEND" in chunks
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), dtype=np.uint8).reshape(60, 1 + 40 * 3)
    assert (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(60, 40, 3), grid.permute(1, 2, 0).numpy())
    assert encode_png(grid[:1])[25] == 0  # colour type of a one-channel image: greyscale
    with pytest.raises(ValueError):
        encode_png(grid.float())


@pytest.mark.gpu
def test_checkpoint_resume(tmp_path, monkeypatch):
    """Trainer.save_checkpoint / load_checkpoint: a run resumed in a fresh Trainer continues with the losses of the run that
    never stopped (both networks, EMA + counter, FlatAdam moments and step count, the iteration counter)."""
    from gan_amd.train_celeba import TrainConfig, Trainer

    dev = torch.device("cuda:0")
    real_randn = torch.randn
    state = {"n": 0}

    def seq_randn(*shape, **kw):  # the k-th draw of a run is the same tensor in every run
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        g = torch.Generator(device="cpu").manual_seed(1000 + state["n"])
        state["n"] += 1
        return real_randn(*shape, generator=g).to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)

    def make():
        torch.manual_seed(21)
        cfg = TrainConfig(model="stylegan2", img_size=32, gen_kwargs={"base_dim": 64}, disc_kwargs={"base_dim": 16}, method="hinge",
                          regularizer="r1", batch_size=8, optimizer="Adam", optimizer_kwargs={"betas": [0, 0.9]}, lr=2e-4, device="cuda",
                          mixed_precision=False, channels_last=False, hip_graph=False)
        return Trainer(cfg, dev)

    reals = real_randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(2)).to(dev)
    monkeypatch.setattr(torch, "randn", seq_randn)
    try:
        a = make()
        for _ in range(2):
            a.train_step(lambda: reals)
        a.save_checkpoint(str(tmp_path / "ck.pt"))
        saved = torch.load(str(tmp_path / "ck.pt"), map_location=dev, weights_only=True)
        mark = state["n"]
        want = [a.train_step(lambda: reals) for _ in range(2)]
        b = make()
        b.load_checkpoint(str(tmp_path / "ck.pt"))
        assert b.step == 2 and b.gen_ema._count == 2

        def flat(obj, prefix=""):
            if isinstance(obj, dict):
                for k, v in obj.items():
                    yield from flat(v, f"{prefix}/{k}")
            elif isinstance(obj, (list, tuple)):
                for i, v in enumerate(obj):
                    yield from flat(v, f"{prefix}/{i}")
            else:
                yield prefix, obj

        restored = dict(flat(b.state_dict()))
        for key, v in flat(saved):  # the restored trainer holds exactly what was saved, bit for bit
            r = restored[key]
            assert (torch.equal(r.cpu(), v.cpu()) if isinstance(v, torch.Tensor) else r == v), key
        state["n"] = mark
        got = [b.train_step(lambda: reals) for _ in range(2)]
    finally:
        monkeypatch.setattr(torch, "randn", real_randn)
    # the continued run tracks the one that never stopped; not bit for bit: the fp32 weight-gradient kernel adds its split-K
    # partials with atomics, and hinge + R1 + Adam(beta1 = 0) amplifies the last bit (1e-3 after four steps)
    for w_, g_ in zip(want, got):
        for k in ("loss/d", "loss/g"):
            assert abs(float(w_[k]) - float(g_[k])) <= 1e-3 * max(1.0, abs(float(w_[k]))), (k, float(w_[k]), float(g_[k]))
    for p, q in zip(a.gen.parameters(), b.gen.parameters()):
        assert (p - q).abs().max() < 1e-4  # (one Adam step moves a weight by lr = 2e-4)
