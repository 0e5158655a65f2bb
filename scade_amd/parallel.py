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
        self._sinks = []            # (net, offset) of the networks attach_grad_sinks() gave a sink
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
        self._sinks = []
        for net in nets:
            ps = net.ordered_params()
            o0 = starts.get(id(ps[0]))
            ok = o0 is not None
            o = o0 if ok else 0
            for p in ps:
                ok = ok and starts.get(id(p)) == o
                o += p.numel()
            net._grad_sink = self.grad[o0:o0 + ops.N_PARAM_FLOATS] if ok else None
            net._sink_fresh = False
            if ok:
                self._sinks.append((net, o0))

    def begin_step(self, zero_outside_sinks: bool = True):
        """Start of a train step, instead of ``zero_grad()``: a network with a gradient sink gets its
        gradient WRITTEN into the sink by the first fused backward of the step (ops.mlp_bwd*(out=sink)), so
        only what lies outside the sinks is zero-filled here (``zero_outside_sinks=False``: not even that - the
        caller's loss operator writes those rows itself).  ``end_backward()`` zero-fills the sink of a
        network whose backward did not run this step."""
        if not self._sinks:
            return self.zero_grad()
        if not zero_outside_sinks:
            for net, _ in self._sinks:
                net._sink_fresh = True
            return
        from . import ops
        pos = 0
        for net, o0 in sorted(self._sinks, key=lambda t: t[1]):
            if o0 > pos:
                self.grad[pos:o0].zero_()
            net._sink_fresh = True
            pos = o0 + ops.N_PARAM_FLOATS
        if pos < self.numel:
            self.grad[pos:].zero_()

    def end_backward(self):
        for net, o0 in self._sinks:
            if getattr(net, "_sink_fresh", False):
                net._grad_sink.zero_()
                net._sink_fresh = False

    def zero_grad(self):
        self.grad.zero_()
        for net, _ in self._sinks:
            net._sink_fresh = False
        o = 0
        for p in self.params:      # re-attach if someone did zero_grad(set_to_none=True)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + n].view(p.shape)
            o += n

    def segment(self, start: int, numel: int) -> "FlatSegment":
        """A contiguous slice of the bucket with the (data, grad, numel, zero_grad) surface FusedAdam
        needs: several optimizers (different learning rates) then share ONE gradient bucket and ONE
        all-reduce."""
        return FlatSegment(self, start, numel)

    def allreduce_grads(self, group=None, force: bool = False) -> float:
        """Sum-all-reduce the gradient bucket; returns the factor the optimizer must apply
        (1/world) so that equal shards with local-mean losses reproduce the global mean.
        (``Trainer`` weights every rank's loss by n_local/N_total instead and applies no factor, which
        is also right for uneven shards.)
        ``force``: issue the collective even on a one-rank group (self-tests of the RCCL path)."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1 and not force:
            return 1.0
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world

    def allreduce_grads_async(self, pieces, group=None, force: bool = False):
        """Sum-all-reduce the bucket as several contiguous pieces ``[(start, numel, stream | None)]``,
        each issued asynchronously behind the work already enqueued on ITS stream (None = the current
        one): a piece whose producer chain ends early starts its collective while the other chains
        still compute.  Returns the work handles (``wait_all`` makes the current stream wait)."""
        if not (dist.is_available() and dist.is_initialized()):
            return []
        if dist.get_world_size(group) == 1 and not force:
            return []
        works = []
        for start, numel, st in pieces:
            if numel <= 0:
                continue
            g = self.grad[start:start + numel]
            if st is not None and g.is_cuda:
                with torch.cuda.stream(st):
                    works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
            else:
                works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
        return works

    @staticmethod
    def wait_all(works):
        for w in works:
            w.wait()

    def broadcast_params(self, src: int = 0, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.data, src=src, group=group)


class FlatSegment:
    """View of ``FlatParams`` rows [start, start+numel): what one FusedAdam updates."""

    def __init__(self, parent: FlatParams, start: int, numel: int):
        if start < 0 or numel <= 0 or start + numel > parent.numel:
            raise ValueError("FlatSegment: range outside the bucket")
        self.parent, self.start, self.numel = parent, start, numel
        self.data = parent.data[start:start + numel]
        self.grad = parent.grad[start:start + numel]

    def zero_grad(self):
        self.grad.zero_()


def batch_share(n_local: int, n_total: Optional[int] = None, group=None) -> float:
    """Weight of this rank's shard in a mean over the GLOBAL batch: n_local / N_total.  With the
    share folded into every rank's loss, a plain sum-all-reduce of the gradients is the gradient of
    the global mean for even AND uneven shards.  ``n_total=None`` assumes equal shards (1/world)
    and costs no communication."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0
    if n_total is None:
        return 1.0 / dist.get_world_size(group)
    return float(n_local) / float(n_total)


def combine_shard_means(means: torch.Tensor, n_local: int, group=None, n_total: Optional[int] = None):
    """In place: per-shard means over ``n_local`` rays -> the mean over the rays of ALL shards
    (one sum-all-reduce of the n_local/N_total-weighted means).  Returns (n_local / N_total, world).
    Single process: identity, (1.0, 1).  ``n_total`` (the global ray count, known to whoever
    sharded the batch) saves the extra all-reduce + host read that finds it, which also makes the
    exchange capturable in a HIP graph."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0, 1
    world = dist.get_world_size(group)
    if n_total is None:
        n = torch.tensor([float(n_local)], device=means.device, dtype=torch.float64)
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
        n_total = float(n.item())
    share = float(n_local) / float(n_total)
    means.mul_(share)
    dist.all_reduce(means, op=dist.ReduceOp.SUM, group=group)
    return share, world


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Rows of a tensor that was split with ``shard_range(n_total, rank, world)`` -> all ``n_total`` rows on
    EVERY rank (one all-gather of equal, zero-padded pieces: the collective needs equal sizes, shards differ
    by at most one row)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_total + world - 1) // world
    a, b = shard_range(n_total, rank, world)
    assert local.shape[0] == b - a, (local.shape, a, b)
    piece = local.new_zeros((per,) + tuple(local.shape[1:]))
    piece[:b - a] = local
    pieces = [torch.empty_like(piece) for _ in range(world)]
    dist.all_gather(pieces, piece, group=group)
    parts = []
    for r in range(world):
        ra, rb = shard_range(n_total, r, world)
        parts.append(pieces[r][:rb - ra])
    return torch.cat(parts, 0)


IMAGE_KEYS = ("rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "disp0", "acc0", "depth0", "z_std")


def render_rays_sharded(rays_flat: torch.Tensor, render_fn, group=None, keys: Optional[Sequence[str]] = IMAGE_KEYS):
    """Test render over the ranks (rays are independent units): this rank renders its
    contiguous share of the ray rows with ``render_fn(rows) -> dict`` and the per-ray maps are all-gathered,
    so every rank ends up with the whole image.  ``keys`` limits what travels (default: the per-pixel maps the
    eval loop reads; ``None`` = every key, including the [N,192] sample arrays).  Rays never interact, so the
    result is bit-identical to the single-process render."""
    n = rays_flat.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return render_fn(rays_flat)
    a, b = shard_range(n, dist.get_rank(group), dist.get_world_size(group))
    ret = render_fn(rays_flat[a:b])
    return {k: gather_rows(v.contiguous(), n, group) for k, v in ret.items() if keys is None or k in keys}


def seed_rank_streams(seed: int, group=None) -> int:
    """Per-rank distinct jitter / u streams (identical weight init on all ranks, distinct
    random draws): seeds the CPU and device generators of this process with ``seed + rank``.  Call it AFTER the
    networks are built (their init wants the SAME seed everywhere, or ``FlatParams.broadcast``); draws that
    must be common to all ranks (sample_pdf_joint's u) go through ``shared_uniform``."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    torch.manual_seed(seed + rank)
    return seed + rank


def shared_uniform(shape, device, generator=None, src: int = 0, group=None) -> torch.Tensor:
    """One uniform draw shared by all ranks (sample_pdf_joint draws ONE u[S] for the whole batch,
    helpers:452-453): drawn on ``src`` and broadcast."""
    u = torch.rand(shape, device=device, generator=generator)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(u, src=src, group=group)
    return u


def staircase_lr(lr0: float, decay_rate: float, decay_step: int, it: int) -> float:
    """train_utils/hyperparameter_update.py:8-13 / run_scade_scannet.py:988-991; ``it`` is the
    reference's loop index i, which counts from 1 (:899-900: start = global_step + 1)."""
    return lr0 * (decay_rate ** (it // decay_step))
