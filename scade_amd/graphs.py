"""Whole train step (zero_grad, render, 3-term loss, backward, Adam) captured in ONE HIP graph.

The eager step costs ~1.7 ms of host time (about sixty launches plus the autograd engine), which
is the limit once the device work is short: a 128-ray shard of a strongly-scaled batch, or the
bf16 path at 1024 rays.  Captured, the step is one ``graph.replay()``.

The Trainer's two-stream backward is captured as a fork/join inside the graph, and so is the RCCL
all-reduce of the gradient bucket when the rays are sharded over ranks (every rank captures and
replays the same sequence).  Fixed shapes; everything that
changes per step lives on the device: inputs in static buffers, Adam's step count / staircase
learning rate / bias corrections in ``FusedAdam.state``."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


class GraphedTrainer:
    def __init__(self, trainer, n_rays: int, n_hyp: int, inject_draws: bool = False,
                 force_allreduce: bool = False):
        tr = self.tr = trainer
        if tr.sharded and tr.cfg["joint"]:
            raise NotImplementedError("GraphedTrainer: the sharded is_joint exchange reads a host scalar")
        self.force_allreduce = force_allreduce
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        dev = tr.flat.data.device
        c = tr.cfg
        self.rays = torch.zeros(n_rays, 11, device=dev)
        self.tgt = torch.zeros(n_rays, 3, device=dev)
        self.hyp = torch.zeros(n_hyp, n_rays, 1, device=dev)
        self.draws = None
        if inject_draws:
            self.draws = (torch.zeros(n_rays, c["Ns"], device=dev), torch.zeros(n_rays, c["Ni"], device=dev),
                          torch.zeros(n_rays, c["Ni"], device=dev))
        tr.opt.use_device_state(c["rate"], c["step"], grad_scale=1.0 / world)
        tr.opt_ss.use_device_state(grad_scale=1.0 / world)
        self.graph = None
        self.loss = None
        self._captured_ss = None     # whether the captured graph contains the scale/shift update

    def _body(self):
        tr = self.tr
        tr.opt.zero_grad()
        tr.opt_ss.zero_grad()
        kw = {}
        if self.draws is not None:
            kw = dict(t_rand=self.draws[0], u_coarse=self.draws[1], cached_u=self.draws[2])
        loss, _ = tr.forward_loss(self.rays, self.tgt, self.hyp, **kw)
        loss.backward()
        tr.flat.allreduce_grads(force=self.force_allreduce)       # 1/world lives in the Adam state
        tr.flat_ss.allreduce_grads(force=self.force_allreduce)
        tr.opt.step_dev()
        if tr.it < tr.cfg["freeze_ss"]:
            tr.opt_ss.step_dev()
        return loss.detach()

    def _capture(self):
        tr = self.tr
        # warm-up (first-call attribute setup, allocator) on a side stream, then roll the state back
        keep = [t.clone() for t in (tr.flat.data, tr.opt.exp_avg, tr.opt.exp_avg_sq, tr.opt.state,
                                    tr.flat_ss.data, tr.opt_ss.exp_avg, tr.opt_ss.exp_avg_sq, tr.opt_ss.state)]
        steps = (tr.opt.steps, tr.opt_ss.steps)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        for dst, src in zip((tr.flat.data, tr.opt.exp_avg, tr.opt.exp_avg_sq, tr.opt.state,
                             tr.flat_ss.data, tr.opt_ss.exp_avg, tr.opt_ss.exp_avg_sq, tr.opt_ss.state), keep):
            dst.copy_(src)
        tr.opt.steps, tr.opt_ss.steps = steps
        ops.PARAM_EPOCH += 1
        self._captured_ss = tr.it < tr.cfg["freeze_ss"]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        # the capture itself executed nothing; undo its host-side bookkeeping
        tr.opt.steps, tr.opt_ss.steps = steps

    def step(self, rays, target_s, target_hyp, t_rand=None, u_coarse=None, cached_u=None):
        """One optimisation step; returns the loss tensor of the step (a static buffer)."""
        self.rays.copy_(rays)
        self.tgt.copy_(target_s)
        self.hyp.copy_(target_hyp)
        if self.draws is not None:
            self.draws[0].copy_(t_rand)
            self.draws[1].copy_(u_coarse)
            self.draws[2].copy_(cached_u)
        tr = self.tr
        with_ss = tr.it < tr.cfg["freeze_ss"]
        if self.graph is None or with_ss != self._captured_ss:
            self._capture()               # first step, or the scale/shift freeze point (:996) was crossed
        self.graph.replay()
        tr.it += 1
        tr.opt.steps += 1
        if with_ss:
            tr.opt_ss.steps += 1
        ops.PARAM_EPOCH += 1          # parameters changed behind the module caches' back
        return self.loss


class GraphedRender:
    """``render_rays`` (inference: no_grad, perturb = 0) of a fixed-size ray batch captured in one
    HIP graph: the ~20 launches of a render step replay with a single call, which is what bounds
    small batches (interactive rendering, the per-GPU shard of a strongly-scaled batch).  The
    outputs are static tensors that the next replay overwrites - clone what must survive."""

    def __init__(self, n_rays: int, network_fn, network_query_fn, N_samples: int, N_importance: int,
                 network_fine, device=None, **render_kw):
        from . import rendering as R
        dev = device if device is not None else next(network_fn.parameters()).device
        self.rays = torch.zeros(n_rays, 11, device=dev)
        self._call = lambda: R.render_rays(self.rays, True, network_fn, network_query_fn, N_samples,
                                           N_importance=N_importance, network_fine=network_fine, perturb=0.,
                                           **render_kw)
        self._nets = (network_fn, network_fine)
        self._epoch = None
        self.graph = None
        self.out = None

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                      # first-launch setup + weight packs outside the capture
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self._call()
        self._epoch = self._weights_key()

    def _weights_key(self):
        return (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for n in self._nets for p in n.parameters())

    def __call__(self, rays):
        """rays [n_rays, >= 11] -> the dict of render_rays (static tensors)."""
        if self.graph is None or self._epoch != self._weights_key():
            # the packed weight blobs are baked into the graph as of capture time: re-capture after
            # any parameter update
            self.rays.copy_(rays[:, :11])
            self._capture()
        else:
            self.rays.copy_(rays[:, :11])
        self.graph.replay()
        return self.out
