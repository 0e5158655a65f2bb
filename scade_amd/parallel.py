"""Ray-parallel data parallelism for the SCADE train step: one process per GPU, rays
sharded, both NeRFs replicated, ONE RCCL all-reduce of a flat fp32 gradient bucket per
step over xGMI, then the identical fused Adam step on every rank.

Replaces the reference's single-process nn.DataParallel over the MLP sample batch
(run_scade_scannet.py:438, :455, :466), which re-broadcasts 2.36 MB of weights and
scatters/gathers the [P,60] embedding on every network call.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of n rays; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(rays, target_s, target_h, rank: int, world: int, mask=None):
    """Slice a global batch: rays [N,11], target_s [N,3], target_h [K,N,1] (dim 1), mask [N]."""
    a, b = shard_range(rays.shape[0], rank, world)
    out = [rays[a:b], target_s[a:b], None if target_h is None else target_h[:, a:b]]
    if mask is not None:
        out.append(mask[a:b])
    return tuple(out)


class FlatParams:
    """Re-homes a list of tensors (nn.Parameters) into ONE flat fp32 buffer and their
    gradients into a second one, so the all-reduce and the optimizer step are single
    operations on contiguous memory (no per-tensor launches, no bucket copies):
    ``p.data`` and ``p.grad`` become views; autograd accumulates into ``p.grad`` in place."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params: List[torch.Tensor] = [p for p in params]
        if not self.params:
            raise ValueError("FlatParams: no parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.data = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        o = 0
        for p in self.params:
            n = p.numel()
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatParams: all tensors must be fp32 on one device")
            self.data[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            o += n

    def attach_grad_sinks(self, nets):
        """Give every NeRF whose 24 parameters sit consecutively (in kernel order) in this buffer a
        direct gradient sink: its fused backward then adds its flat gradient with one kernel
        instead of handing 24 tensors to autograd's per-tensor accumulation."""
        from . import ops
        starts, o = {}, 0
        for p in self.params:
            starts[id(p)] = o
            o += p.numel()
        for net in nets:
            ps = net.ordered_params()
            o0 = starts.get(id(ps[0]))
            ok = o0 is not None
            o = o0 if ok else 0
            for p in ps:
                ok = ok and starts.get(id(p)) == o
                o += p.numel()
            net._grad_sink = self.grad[o0:o0 + ops.N_PARAM_FLOATS] if ok else None

    def zero_grad(self):
        self.grad.zero_()
        o = 0
        for p in self.params:      # re-attach if someone did zero_grad(set_to_none=True)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + n].view(p.shape)
            o += n

    def allreduce_grads(self, group=None, force: bool = False) -> float:
        """Sum-all-reduce the gradient bucket; returns the factor the optimizer must apply
        (1/world) so that equal shards with local-mean losses reproduce the global mean.
        ``force``: issue the collective even on a one-rank group (self-tests of the RCCL path)."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1 and not force:
            return 1.0
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world

    def broadcast_params(self, src: int = 0, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.data, src=src, group=group)


def combine_shard_means(means: torch.Tensor, n_local: int, group=None):
    """In place: per-shard means over ``n_local`` rays -> the mean over the rays of ALL shards
    (one sum-all-reduce of the n_local/N_total-weighted means).  Returns (n_local / N_total, world).
    Single process: identity, (1.0, 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0, 1
    world = dist.get_world_size(group)
    n = torch.tensor([float(n_local)], device=means.device, dtype=torch.float64)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    share = float(n_local) / float(n.item())
    means.mul_(share)
    dist.all_reduce(means, op=dist.ReduceOp.SUM, group=group)
    return share, world


def shared_uniform(shape, device, generator=None, src: int = 0, group=None) -> torch.Tensor:
    """One uniform draw shared by all ranks (sample_pdf_joint draws ONE u[S] for the whole batch,
    helpers:452-453): drawn on ``src`` and broadcast."""
    u = torch.rand(shape, device=device, generator=generator)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(u, src=src, group=group)
    return u


def staircase_lr(lr0: float, decay_rate: float, decay_step: int, it: int) -> float:
    """train_utils/hyperparameter_update.py:8-13 / run_scade_scannet.py:988-991."""
    return lr0 * (decay_rate ** (it // decay_step))
