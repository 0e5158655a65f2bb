"""Whole train step (zero_grad, render, 3-term loss, backward, Adam) captured in ONE HIP graph.

The eager step costs ~0.7-0.8 ms of host time (about thirty launches plus the autograd engine; 1.7 ms before
round 3's launch fusion), which is the limit once the device work is short: a 128-ray shard of a strongly-scaled
batch, or the 16-bit paths at 1024 rays.  Captured, the step is one ``graph.replay()``; the training loop
(driver.train_scene) assembles each batch straight into the static buffers with one launch (``step_staged``).

The Trainer's two-stream backward is captured as a fork/join inside the graph, and so is the RCCL
all-reduce of the gradient bucket when the rays are sharded over ranks (every rank captures and
replays the same sequence).  Fixed shapes; everything that
changes per step lives on the device: inputs in static buffers, Adam's step count / staircase
learning rate / bias corrections in ``FusedAdam.state``."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops
from .optim import adam_step_pair


CAPTURE_ERROR_MODE = "thread_local"


def _capture_mode():
    """Capture-error mode of every stream capture of this module: "thread_local".  torch's default, "global",
    makes a HIP call that is illegal during capture an error in EVERY thread of the process - including the NCCL
    watchdog thread of a torch.distributed process group, which polls the events of earlier (eager) collectives
    with hipEventQuery.  When such a poll lands inside a capture window the watchdog throws and the process
    aborts (ProcessGroupNCCL.cpp, Watchdog::run): the "RCCL teardown abort" of round 2, seen about once in forty
    to sixty runs and never in teardown (tools/probe_rccl_teardown.py, profiles/r03_rccl_teardown.txt).  The
    capturing thread's own calls stay checked.  (``graphs.CAPTURE_ERROR_MODE = "global"`` restores the old behaviour:
    the probe tool's variant G.)"""
    return CAPTURE_ERROR_MODE


class GraphedTrainer:
    """``n_total``: rays of the whole (all-rank) batch when the shards are uneven (default: equal
    shards, n_rays x world); it fixes this rank's loss weight n_rays / n_total at capture time.
    ``with_mask``: keep a static mask buffer [n_rays] and pass ``mask=`` to every step."""

    def __init__(self, trainer, n_rays: int, n_hyp: int, inject_draws: bool = False,
                 force_allreduce: bool = False, n_total=None, with_mask: bool = False):
        tr = self.tr = trainer
        tr.force_allreduce = tr.force_allreduce or force_allreduce
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.n_total = n_total if n_total is not None else n_rays * world
        dev = tr.bucket.data.device
        c = tr.cfg
        self.rays = torch.zeros(n_rays, 11, device=dev)
        self.tgt = torch.zeros(n_rays, 3, device=dev)
        self.hyp = torch.zeros(n_hyp, n_rays, 1, device=dev)
        # which image's scale / shift the batch belongs to (:951-954): a device index, so that the
        # captured graph gathers (and updates) the right row every step
        self.img_i = torch.zeros(1, device=dev, dtype=torch.long)
        self.mask = torch.ones(n_rays, device=dev) if with_mask else None
        self.draws = None
        if inject_draws:
            self.draws = (torch.zeros(n_rays, c["Ns"], device=dev), torch.zeros(n_rays, c["Ni"], device=dev),
                          torch.zeros(n_rays, c["Ni"], device=dev))
        # the staircase runs on the reference's loop index (:899-900): a resumed Trainer (start_iter = N,
        # optimizer state not restored, :480) has taken opt.steps = 0 steps but stands at iteration N
        tr.opt.use_device_state(c["rate"], c["step"], iter_offset=tr.it - tr.opt.steps)
        tr.opt_ss.use_device_state()
        # in-kernel draws: the step index must live on the device too (state[13] = steps taken BEFORE the step in flight:
        # the tick that opens the step, ops.stage_inputs below, records it;
        # resumed runs add their offset through the seed so that the streams do not repeat)
        # Kept HERE, not on the Trainer: eager Trainer.step calls on the same trainer keep drawing by their host
        # step index, and a second GraphedTrainer on it does not apply the offset twice.
        self._resume_offset = tr.it - tr.opt.steps
        self._step_dev = tr.opt.state[13:14]
        # in-kernel draws: the coarse samples of a step (z_vals, positions, both samplers' draws) are computed by the
        # launch that opens it (ops.stage_inputs / ops.ResidentBatchGather ``points=``), not by a launch of the
        # captured step: one launch less per iteration, same kernel body, same bits.  ``coarse_pre = None`` (set it
        # before the first step) keeps scade_ray_points_draw inside the graph.
        # ... and so are the step's weight packs (ops.StepPacks: re-packed in place from what the previous replay's
        # optimizer launch left); ``packs = None`` keeps scade_mlp_pack_step inside the graph
        fmt = {"f32": "f32", "bf16": "bf16", "bf16-s8": "bf16", "f16": "f16", "f16x3": "f16x3"}.get(tr.coarse.train_precision)
        self.packs = ops.StepPacks([tr.coarse, tr.fine], fmt) if (fmt and tr.fine.train_precision == tr.coarse.train_precision) else None
        # does the launch in front of every replay pack?  ``step()`` does; a caller that drives ``step_staged()`` with its
        # own opening launch sets this when that launch packs (driver.train_scene: ResidentBatchGather(packs=...)) - a
        # replay of a graph captured WITHOUT its own pack behind an opening launch that does not pack would run on the
        # previous step's blobs, so the flag is part of what a capture is valid for
        self.opening_packs = False
        self._cap_packs = None
        self.coarse_pre = None
        if self.draws is None and not c["joint"]:
            self.coarse_pre = ops.CoarsePoints(n_rays, c["Ns"], c["Ni"], c["lindisp"], dev, key=self._points_key,
                                               step=lambda: tr.opt.steps)
        self.graph = None
        self.loss = None
        self.terms = None
        self._captured = None     # (scale/shift update in the graph?, carving term in the graph?)
        self._key = None          # Philox key baked into the captured launches (None: draws are injected)

    def _points_key(self):
        return (self.tr.draw_key() + self._resume_offset * 0x2545F4914F6CDD1D) & (2 ** 64 - 1)

    def _body(self):
        tr = self.tr
        tr.begin()
        kw = {}
        if self.draws is not None:
            kw = dict(t_rand=self.draws[0], u_coarse=self.draws[1], cached_u=self.draws[2])
        elif self.coarse_pre is not None:
            self._key = tr.draw_key()          # (not baked into the graph any more; kept for the re-capture rule)
            kw = dict(_coarse_pre=self.coarse_pre)
        else:
            # the Philox key is a kernel ARGUMENT of the captured launches: recorded, and step() re-captures when
            # Trainer.reseed_draws() / a new torch seed changed it (eager and graphed steps stay on one stream)
            self._key = tr.draw_key()
            key = (self._key + self._resume_offset * 0x2545F4914F6CDD1D) & (2 ** 64 - 1)
            kw = dict(draws=ops.Draws(key, 0, self._step_dev))
        loss, aux = tr.forward_loss(self.rays, self.tgt, self.hyp, img_i=self.img_i, mask=self.mask,
                                    n_total=self.n_total, **kw)
        if tr._unit_loss_ready:
            tr._unit_loss_ready = False
            tr.flat_ss.grad.zero_()
        tr.backward(loss, defer_reduce=True)
        tr.bucket.end_backward()
        tr.reduce_grads()
        tr.finish(dev=True)
        # the three loss terms of the step (:968-983) as static tensors too: the loop's log line reads them
        self.terms = (aux["img_loss"], aux["carve"], aux["img_loss0"])
        return aux["loss_report"]

    def _state(self):
        tr = self.tr
        return (tr.bucket.data, tr.opt.exp_avg, tr.opt.exp_avg_sq, tr.opt.state,
                tr.opt_ss.exp_avg, tr.opt_ss.exp_avg_sq, tr.opt_ss.state)

    def _capture(self):
        tr = self.tr
        # warm-up (first-call attribute setup, allocator) on a side stream, then roll the state back
        keep = [t.clone() for t in self._state()]
        steps = (tr.opt.steps, tr.opt_ss.steps)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        for dst, src in zip(self._state(), keep):
            dst.copy_(src)
        tr.opt.steps, tr.opt_ss.steps = steps
        ops.PARAM_EPOCH += 1
        self._cap_packs = self.packs is not None and self.opening_packs
        if self._cap_packs:
            # the blobs from the rolled-back parameters, outside the capture, marked fresh: the captured body's own pack
            # finds nothing to do and the graph starts at the first MLP launch; every replay's blobs come from the launch
            # in front of it (``packs=`` of ops.stage_inputs / ops.ResidentBatchGather)
            self.packs.prepare()
        self._captured = (tr.scaleshift_active(), tr.carving_active())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.loss = self._body()
        # the capture itself executed nothing; undo its host-side bookkeeping
        tr.opt.steps, tr.opt_ss.steps = steps

    def step(self, rays, target_s, target_hyp, t_rand=None, u_coarse=None, cached_u=None, img_i=0, mask=None):
        """One optimisation step; returns the loss tensor of the step (a static buffer)."""
        tr = self.tr
        if (mask is None) != (self.mask is None):
            raise ValueError("GraphedTrainer.step: pass mask= exactly when built with with_mask=True")
        # the step's inputs -> the static buffers, in ONE launch (ops.stage_inputs; a caller that assembled its
        # batch in ``self.rays`` / ``self.tgt`` / ``self.hyp`` directly pays nothing)
        pairs = [(rays, self.rays), (target_s, self.tgt), (target_hyp, self.hyp)]
        scalar = None
        if torch.is_tensor(img_i):
            pairs.append((img_i.reshape(1), self.img_i))
        else:
            if not 0 <= int(img_i) < tr.n_images:
                raise IndexError(f"GraphedTrainer.step: img_i {img_i} outside [0, {tr.n_images})")
            scalar = (self.img_i, int(img_i))
        if mask is not None:
            pairs.append((mask, self.mask))
        if self.draws is not None:
            pairs += [(t_rand, self.draws[0]), (u_coarse, self.draws[1]), (cached_u, self.draws[2])]
        # ... and the optimizers' device-resident step scalars advance in the same launch (no tick launch in the graph)
        self.opening_packs = self.packs is not None
        if self.packs is not None and self.graph is None:
            self.packs.prepare()          # (the blobs must exist before the first opening launch re-packs them)
        ops.stage_inputs(pairs, scalar, tick=self.tick_states(), points=self.coarse_pre,
                         rays=rays if self.coarse_pre is not None else None, packs=self.packs)
        return self.step_staged()

    def tick_states(self):
        """The device-resident optimizer scalars the launch in FRONT of a replay advances (``ops.stage_inputs`` /
        ``ops.gather_batch`` ``tick=``): the networks' Adam and, until the freeze point (:996), the scale / shift one."""
        tr = self.tr
        return [tr.opt.state, tr.opt_ss.state if tr.scaleshift_active() else None]

    def step_staged(self):
        """The step on what the static buffers hold NOW: the caller has written ``rays`` / ``tgt`` / ``hyp`` /
        ``img_i`` (/ ``mask`` / ``draws``) itself and advanced ``tick_states()`` in that launch - the training
        loop's fused batch gather (``ops.ResidentBatchGather``) does.  One ``graph.replay()``."""
        tr = self.tr
        with_ss = tr.scaleshift_active()
        if self.graph is None or (with_ss, tr.carving_active()) != self._captured or \
                self._cap_packs != (self.packs is not None and self.opening_packs) or \
                (self._key is not None and self._key != tr.draw_key()):
            self._capture()      # first step, the warm-start (:973) / scale-shift freeze point (:996) was crossed, or
            #                      the draws were re-seeded (the key is baked into the captured launches)
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
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()), torch.no_grad():
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
