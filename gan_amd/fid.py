"""FID (reference fid.py:12-39): the same class and methods -- `FID(device).compute_stats(closure)` -> dict(mean, cov) over
10 000 feature vectors, static `fid_score(stats1, stats2)` with the eigenvalue form of the Frechet distance (fid.py:33-39).

What differs, and why:
* the feature network.  The reference takes `pytorch_fid.inception.InceptionV3()` (fid.py:4,15), whose weights are a download;
  neither the package nor the weights exist in this image, so the default constructor raises ImportError with that message
  and a caller passes `feature_net` (any module mapping images to (N, F) features, or to a list whose first element
  flattens to that, as InceptionV3 does).  The statistics and the score do not depend on which network it is.
* data parallel: with a process group up every rank evaluates its share of the samples and the features are all-gathered
  (SURVEY.md 8 f3 "distributed feature gather"); every rank returns the same statistics.
Not part of the training hot path: plain library calls (the covariance is one GEMM, the score one eigenvalue problem)."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
from torch import Tensor, nn


class FID(nn.Module):
    def __init__(self, device: str, feature_net: Optional[nn.Module] = None, n_samples: int = 10_000) -> None:
        super().__init__()
        if feature_net is None:
            try:
                from pytorch_fid.inception import InceptionV3  # fid.py:4
            except ImportError as exc:  # not installable here: say so instead of guessing
                raise ImportError("FID needs pytorch_fid's InceptionV3 (and its downloaded weights) as the reference does, "
                                  "or a feature_net of your own: FID(device, feature_net=...)") from exc
            feature_net = InceptionV3()
        self.inception = feature_net.to(device).eval()
        self.device = device
        self.n_samples = n_samples

    def _features(self, imgs: Tensor) -> Tensor:
        out = self.inception(imgs)
        if isinstance(out, (list, tuple)):
            out = out[0]
        return out.flatten(1).float()

    def compute_stats(self, closure: Callable[[], Tensor]) -> Dict[str, Tensor]:
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        n = self.n_samples
        mine = -(-n // world)  # every rank fills the same number of rows: all_gather needs equal shapes
        feats: Optional[Tensor] = None
        i = 0
        while i < mine:
            with torch.no_grad():
                out = self._features(closure())
            if feats is None:
                feats = torch.empty(mine, out.shape[1], device=out.device)
            j = min(i + out.shape[0], mine)
            feats[i:j] = out[: j - i]
            i = j
        if world > 1:
            parts = [torch.empty_like(feats) for _ in range(world)]
            dist.all_gather(parts, feats)
            feats = torch.cat(parts, 0)
        feats = feats[:n]
        return dict(mean=feats.mean(0).cpu(), cov=feats.T.cov().cpu())

    @staticmethod
    def fid_score(stats1: Dict[str, Tensor], stats2: Dict[str, Tensor]) -> Tensor:
        a = (stats1["mean"] - stats2["mean"]).square().sum(dim=-1)
        b = stats1["cov"].trace() + stats2["cov"].trace()
        c = torch.linalg.eigvals(stats1["cov"] @ stats2["cov"]).sqrt().real.sum(dim=-1)
        return a + b - 2 * c
