"""Batch data parallelism for the train step: one process per GPU, gradients averaged with bucketed
all-reduces on RCCL (torch.distributed backend "nccl" on ROCm) over xGMI, launched from autograd hooks
so they overlap the rest of the backward pass.

The reference is single-GPU (README.md:39); what pins the semantics is that every loss and regularizer
is a batch mean (losses.py:57,67,96,102), so equal shards + a gradient MEAN reproduce the global-batch
gradient.  Three cross-sample couplings:
  * MinibatchStdDev groups samples {j + k*N/4} (progressive_gan.py:27): `interleaved_shard` gives each
    rank whole groups, ordered so that the local view(4, M/R, ...) rebuilds the global groups.
  * the DCGAN generator's BatchNorm uses per-replica statistics under DP (documented divergence).
  * RNG: parity tests slice one global z / noise draw; throughput runs seed per rank.

Design for MI355X: parameters and gradients live in flat arenas (gan_amd.utils.FlatArena); a bucket is
a contiguous slice of the gradient arena, so an all-reduce needs no gather/scatter copies.  Buckets are
formed in REVERSE parameter order (the order backward produces gradients) with a size chosen for
xGMI's point-to-point links: ~8 MB keeps a ring step per link (~1 MB at 8 GPUs) well above the latency
floor while leaving ~5 buckets per 40 MB network to overlap with the double-backward convolutions.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .utils import FlatArena, arena_of


def init_from_env(device_type: Optional[str] = None) -> Tuple[int, int, torch.device]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract); returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    # GAN_AMD_ONE_GPU=1 (rehearsal aid): every rank uses cuda:0 and the collectives go over gloo, so the
    # multi-process path can be exercised on a single-GPU box; RCCL needs one device per rank
    one_gpu = use_gpu and os.environ.get("GAN_AMD_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    # GAN_AMD_FORCE_ALLREDUCE=1 (test aid): bring the process group up even for one rank, so that the RCCL path,
    # its watchdog thread and the HIP-graph capture meet on a one-GPU box
    if (world > 1 or os.environ.get("GAN_AMD_FORCE_ALLREDUCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if use_gpu and not one_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, device


def _via_host(t: Tensor, group=None) -> bool:
    """gloo with device tensors (the GAN_AMD_ONE_GPU rehearsal: several ranks on one GPU) stages through the host;
    its own device-tensor path hung in the first broadcast on this ROCm build."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: Tensor, group=None, async_op: bool = False):
    """dist.all_reduce(SUM) in place; returns the work handle for async_op on RCCL, None otherwise."""
    if _via_host(t, group):
        c = t.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast(t: Tensor, src: int = 0, group=None) -> None:
    if _via_host(t, group):
        c = t.detach().cpu()
        dist.broadcast(c, src=src, group=group)
        t.copy_(c)
        return
    dist.broadcast(t, src=src, group=group)


def interleaved_shard(global_batch: int, rank: int, world: int, group_size: int = 4) -> Tensor:
    """Row indices of the global batch owned by `rank`, such that MinibatchStdDev on the local shard sees
    exactly the global groups: rank r owns group columns J_r = [r*M/R, (r+1)*M/R), M = N/group_size, and
    the rows {j + k*M : j in J_r, k = 0..G-1} ordered k-major."""
    if global_batch % (group_size * world) != 0:
        raise ValueError(f"global batch {global_batch} must be a multiple of group_size*world = {group_size * world}")
    m = global_batch // group_size
    per = m // world
    cols = torch.arange(rank * per, (rank + 1) * per)
    return (cols.view(1, -1) + torch.arange(group_size).view(-1, 1) * m).reshape(-1)


class GradBucketReducer:
    """Averages a module's gradients across ranks, one async all-reduce per bucket, fired from
    post-accumulate-grad hooks as soon as every gradient of the bucket has been produced.

    usage per backward pass:   reducer.prepare();  loss.backward();  reducer.finish()
    """

    def __init__(self, module: nn.Module, bucket_bytes: int = 8 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # GAN_AMD_FORCE_ALLREDUCE=1 (test aid): issue the bucket all-reduces in a 1-rank group too, so the RCCL
        # calls, their hooks and the stream hand-over run on a single-GPU box
        self._force = dist.is_initialized() and os.environ.get("GAN_AMD_FORCE_ALLREDUCE") == "1"
        self.arena: FlatArena = arena_of(module, with_grad=True)
        self.arena.attach_grads()
        self.buckets: List[Tuple[int, int, List[int]]] = []  # (start, end, param indices)
        self._build(bucket_bytes)
        self._bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self._bucket_of[i] = b
        self._pending: List[int] = []
        self._fired: List[bool] = []
        self._works = []
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.arena.params)]

    def _build(self, bucket_bytes: int) -> None:
        n = len(self.arena.params)
        cap = max(1, bucket_bytes // 4)
        end = self.arena.numel
        idxs: List[int] = []
        for i in range(n - 1, -1, -1):
            idxs.append(i)
            start = self.arena.offsets[i]
            if end - start >= cap or i == 0:
                self.buckets.append((start, end, idxs))
                end, idxs = start, []

    def _make_hook(self, index: int):
        def hook(param):
            if not self._armed:
                return
            b = self._bucket_of[index]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

        return hook

    def _launch(self, b: int) -> None:
        if self._fired[b]:
            return
        self._fired[b] = True
        if self.world == 1 and not self._force:
            return
        start, end, _ = self.buckets[b]
        buf = self.arena.grad[start:end]
        # pre-divide so SUM gives the mean on every backend (gloo has no AVG)
        buf.div_(self.world)
        work = all_reduce_sum(buf, group=self.group, async_op=True)
        if work is not None:
            self._works.append(work)

    def prepare(self) -> None:
        """Arm the hooks for one backward pass.  Parameters that do not require grad (frozen D during
        the generator step, train_celeba.py:140) are not waited for."""
        if not self.arena.intact():
            raise RuntimeError("GradBucketReducer: parameters were moved after flattening; rebuild the reducer")
        for p, o in zip(self.arena.params, self.arena.offsets):
            if p.grad is None or p.grad.data_ptr() != self.arena.grad.data_ptr() + 4 * o:
                self.arena.attach_grads()  # someone called zero_grad(set_to_none=True)
                self.arena.grad.zero_()
                break
        self._pending = [sum(1 for i in idxs if self.arena.params[i].requires_grad) for _, _, idxs in self.buckets]
        self._fired = [False] * len(self.buckets)
        self._works = []
        self._armed = True

    def finish(self) -> None:
        """Reduce whatever the hooks have not (parameters that received no gradient) and wait."""
        self._armed = False
        for b, (_, _, idxs) in enumerate(self.buckets):
            # partially filled or untouched bucket (a parameter got no gradient this pass): reduce it anyway
            # so that every rank issues the same collectives in the same order
            if not self._fired[b] and any(self.arena.params[i].requires_grad for i in idxs):
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []

    def zero_grad(self) -> None:
        self.arena.grad.zero_()

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
