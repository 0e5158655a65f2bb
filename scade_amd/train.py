"""Thin train-step driver reproducing the reference's per-iteration call sequence
(run_scade_scannet.py:951-997) on the HIP operators, one process per GPU:

    target_h = hyp*scale + shift                      :954
    render_hyp -> render_rays(perturb=1)              :963
    loss = mse(rgb) + w*carve(pred_hyp) + mse(rgb0)   :968-983
    backward                                          :985
    [RCCL all-reduce of the flat gradient bucket]     (replaces nn.DataParallel, :438/:455)
    staircase LR, Adam step (+ scale/shift Adam)      :988-997

Not a port of train_nerf (data loading, logging, checkpoints stay with the caller).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from . import rendering as R
from . import run_nerf_helpers as H
from .optim import FusedAdam, adam_step_pair
from .parallel import FlatParams, batch_share, shared_uniform, staircase_lr


def make_scade_nets(device, seed: Optional[int] = None):
    """coarse + fine NeRF in the SCADE configuration (create_nerf, :425-455)."""
    if seed is not None:
        torch.manual_seed(seed)
    mk = lambda: H.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3,
                        input_ch_cam=0, use_viewdirs=True).to(device)
    return mk(), mk()


class Trainer:
    """``mask_mode``: which loss terms a ``mask`` passed to ``step`` multiplies - "scannet"
    (run_scade_scannet.py:968-983: the space-carving loss only) or "wild" (run_scade_wild.py:977-1008:
    both photometric terms as well).  ``allreduce``: "single" = ONE sum-all-reduce of the whole
    gradient bucket per step; "staged" = the same bucket in two pieces WITHOUT giving up the
    joint backward: the two networks share one dgrad launch, then the coarse network's weight gradient + reduce run
    on their own, its piece of the bucket starts its all-reduce right behind them (asynchronously: the collective
    waits for the stream as it stands and the stream goes on), and the fine network's weight gradient - the
    longest kernel of the step - runs on top of it; the fine piece (+ the scale / shift rows) follows its reduce.
    "overlap" (round 2) = two pieces with the coarse stage on a side stream, its all-reduce behind the coarse
    backward CHAIN - which gives up the joint backward (+0.13 / +0.27 ms per step before a byte is on the wire)."""

    def __init__(self, coarse, fine, bb_center, bb_scale, n_images=1, lrate=5e-4, scaleshift_lr=1e-7,
                 space_carving_weight=0.007, N_samples=64, N_importance=128, lrate_decay_rate=0.1,
                 lrate_decay_step=400000, freeze_ss=400000, norm_p=2, space_carving_threshold=0.0,
                 is_joint=False, warm_start_nerf=0, lindisp=False, raw_noise_std=0.0, precision="f32",
                 overlap_coarse=None, mask_mode="scannet", allreduce=None, start_iter=0, fused_loss=True,
                 joint_backward=None):
        dev = next(coarse.parameters()).device
        if mask_mode not in ("scannet", "wild"):
            raise ValueError('Trainer: mask_mode must be "scannet" or "wild"')
        self.coarse, self.fine = coarse, fine
        coarse.train_precision = fine.train_precision = precision      # "f32" (exact) | "f16x3"
        embed_fn, _ = H.get_embedder(9, 0)
        embeddirs_fn, _ = H.get_embedder(0, 0)
        self.query = R.make_network_query_fn(embed_fn, embeddirs_fn, bb_center.to(dev), bb_scale.to(dev))
        # DEPTH_SCALES / DEPTH_SHIFTS, one per training image (:878-888)
        self.n_images = n_images
        self.depth_scales = torch.ones(n_images, 1, device=dev, requires_grad=True)
        self.depth_shifts = torch.zeros(n_images, 1, device=dev, requires_grad=True)
        # ONE gradient bucket: [coarse | fine | scales | shifts] = 2 x 589,700 + 2 n_images floats
        # (SURVEY section 8e); the two optimizers (different learning rates) update segments of it
        self.bucket = FlatParams(list(coarse.parameters()) + list(fine.parameters())
                                 + [self.depth_scales, self.depth_shifts])
        self.bucket.attach_grad_sinks([coarse, fine])
        self.n_net = sum(p.numel() for p in coarse.parameters()) + sum(p.numel() for p in fine.parameters())
        self.n_coarse = sum(p.numel() for p in coarse.parameters())
        self.flat = self.bucket.segment(0, self.n_net)
        self.flat_ss = self.bucket.segment(self.n_net, 2 * n_images)
        self.opt = FusedAdam(self.flat, lr=lrate, betas=(0.9, 0.999))
        self.opt_ss = FusedAdam(self.flat_ss, lr=scaleshift_lr)
        self.cfg = dict(lrate=lrate, rate=lrate_decay_rate, step=lrate_decay_step, w=space_carving_weight,
                        Ns=N_samples, Ni=N_importance, freeze_ss=freeze_ss, norm_p=norm_p,
                        thr=space_carving_threshold, joint=is_joint, warm=warm_start_nerf,
                        lindisp=lindisp, noise=raw_noise_std, mask_mode=mask_mode)
        # ``it`` = optimisation steps taken; the reference's loop index of the NEXT step is it + 1
        # (:899-900: i runs from global_step + 1)
        self.it = start_iter
        # rays are sharded over the ranks of the default process group (one process per GPU)
        self.sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if allreduce is None:
            allreduce = "single"
        if allreduce not in ("single", "overlap", "staged"):
            raise ValueError('Trainer: allreduce must be "single", "staged" or "overlap"')
        self._staged_works, self._staged_done = [], []
        self.allreduce = allreduce
        # coarse stage on a side stream: its backward chain then runs beside the fine one.  Off by default
        # since round 2: with two weight-gradient workgroups per CU, two kernels sharing the chip no longer
        # beat the same two back to back (measured 1024 rays, on / off: exact 7.28 / 7.17 ms, bf16 1.55 /
        # 1.50 ms, f16x3 3.47 / 3.43 ms).  The two-piece overlapped all-reduce needs the side stream and
        # turns it on; ``overlap_coarse=True / False`` forces it.
        if overlap_coarse is None:
            overlap_coarse = allreduce == "overlap"
        self.coarse_stream = torch.cuda.Stream(device=dev) if overlap_coarse and dev.type == "cuda" else None
        self.force_allreduce = False        # self-tests: issue the collective on a one-rank group too
        # the three-term loss as one fused operator (ops.TrainLossFn) instead of the separate public
        # operators (same arithmetic; ``fused_loss=False`` keeps the operator-by-operator path)
        self.fused_loss = fused_loss
        # the MLP backward of both networks as ONE dgrad / weight-gradient / reduce launch each
        # (mlp_bwd.DeferredBackward).  Not with the side stream: there the coarse chain's own launches are the point
        if joint_backward is None:
            joint_backward = True
        self.joint_backward = bool(joint_backward) and self.coarse_stream is None
        self._unit_loss_ready = False
        # fine tail + loss + both tails' backward in one launch (forward_loss) where the step's mode allows it;
        # ``fused_tail_loss = False``: the separate operators (same bits: the equality is a test)
        self.fused_tail_loss = True
        self._draw_key = None               # Philox key of the in-kernel draws: draw_key(), derived at first use
        # d loss / d loss = 1 (see _unit_grad; built here, never inside a graph capture).  Private and IMMUTABLE: the
        # one-launch loss forms bake the factor 1.0 in and recognise this tensor by address and version
        self._one = torch.ones((), device=dev)
        # the weight gradient's reduce rides in the optimizer's launch (ops.step_finish) when no gradient exchange sits
        # between backward and optimizer; ``fused_finish = False`` keeps the separate launches (same bits: a test)
        self.fused_finish = precision in ("f32", "bf16", "bf16-s8", "f16", "f16x3")
        self._pending_reduce = None
        # the loss-scale maxima of the 16-bit backward come out of the fine tail + loss launch (no lp_gmax launch);
        # False: the backward computes them itself (tests compare the two)
        self.tail_gmax = True
        self.bucket.broadcast_params(0)

    def draw_key(self) -> int:
        """Philox key of this rank's in-kernel draws.  Derived at the FIRST step, not at construction, from torch's
        seed of this process at that moment (so ``torch.manual_seed`` / ``parallel.seed_rank_streams`` called after
        the Trainer was built still select the stream, as they did for the ``torch.rand`` path) AND the rank: the
        Philox counter is (draw block, local ray, step), so two ranks with one key would draw the same jitter and
        the same u for their k-th ray (per-rank distinct u / jitter streams)."""
        if self._draw_key is None:
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self._draw_key = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x5CADE
                              + rank * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        return self._draw_key

    def reseed_draws(self):
        """Forget the key: the next step derives it again from torch's current seed."""
        self._draw_key = None

    def _unit_grad(self, loss):
        """d loss / d loss = 1 from a cached tensor (autograd would launch a fill for it every step)."""
        one = getattr(self, "_one", None)
        if one is None or one.shape != loss.shape or one.device != loss.device:
            one = self._one = torch.ones_like(loss)
        return one

    # -- reference loop predicates on i = it + 1 ------------------------------------------------
    def carving_active(self):
        return self.cfg["w"] > 0. and (self.it + 1) > self.cfg["warm"]                        # :973

    def scaleshift_active(self):
        return (self.it + 1) < self.cfg["freeze_ss"]                                          # :996

    def forward_loss(self, rays, target_s, target_hyp, img_i=0, mask=None, n_total=None, **render_kw):
        """-> (loss, aux).  ``loss`` is this rank's term of the GLOBAL-batch loss: every mean over rays
        is weighted by n_local / N_total (``n_total`` = rays of all ranks; None = equal shards), so the
        sum over ranks is the single-process loss and the sum-all-reduced gradient its gradient."""
        c = self.cfg
        # the MFMA weight blobs of both networks (forward + transposed layout), rebuilt after the last
        # optimizer step by ONE launch instead of four
        prec = self.coarse.train_precision
        if (prec == "f32" or prec in ops.LP_FORMATS) and self.fine.train_precision == prec \
                and torch.is_grad_enabled():
            ops.mlp_pack_step([self.coarse, self.fine], "bf16" if prec == "bf16-s8" else prec)
        elif prec in ("f16x3", "f16x3-dgrad") and self.fine.train_precision == prec and torch.is_grad_enabled():
            ops.mlp_pack_step_f16x3([self.coarse, self.fine])
        share = batch_share(rays.shape[0], n_total) if self.sharded else 1.0
        if c["joint"] and self.sharded:
            # the LAST sampler (sample_pdf_joint_return_u, :728) draws ONE u[S] for the whole batch
            # (helpers:498-513): rank 0's draw.  The coarse importance sampler stays per ray (:705).
            render_kw.setdefault("cached_u", shared_uniform((c["Ni"],), rays.device))
        if not any(k in render_kw for k in ("t_rand", "u_coarse", "pytest", "draws", "_coarse_pre")):
            # the step's uniform draws (stratified jitter :564-579, the sample_pdf draws helpers:346-361,
            # :395-410) are made inside the step's first kernel: no generator launch, no jitter tensor.
            # Injected draws (parity tests, the reference's pytest=True streams) take precedence.
            render_kw["draws"] = ops.Draws(self.draw_key(), self.it)
        hyp_per_ray = target_hyp.dim() == 3 and target_hyp.shape[-1] == 1
        # the fine tail, the loss and the backward of both tails as ONE launch (ops.FineTailLossFn) where the step
        # allows it: unit-gradient loss, per-ray hypotheses, no raw noise, coarse stage on this stream
        one_launch = (self.fused_tail_loss and self.fused_loss and not c["joint"] and hyp_per_ray
                      and self._unit_loss_ready and torch.is_grad_enabled() and c["noise"] == 0.
                      and self.coarse_stream is None and "pytest" not in render_kw)
        ret = R.render_rays(rays, True, self.coarse, self.query, c["Ns"], N_importance=c["Ni"],
                            network_fine=self.fine, perturb=1., raw_noise_std=c["noise"],
                            lindisp=c["lindisp"], is_joint=c["joint"], coarse_stream=self.coarse_stream,
                            _stop_before_fine_tail=one_launch, **render_kw)
        if ret.pop("_fine_tail_pending", False):
            self._unit_loss_ready = False
            raw, raw0, u = ret.pop("raw"), ret.pop("raw0"), ret["u"]
            loss, comps, rgb, disp, acc, w, depth, pred, std = ops.FineTailLossFn.apply(
                raw, ret["z_vals"], rays, u, c["Ni"], raw0, ret["z_vals0"], ret["rgb0"].detach(), target_s, target_hyp,
                self.depth_scales, self.depth_shifts, img_i, mask, c["mask_mode"] == "wild", self.carving_active(),
                c["w"], c["thr"], share, self._one,
                # (the formats whose saved dZ rows carry a launch-wide loss scale, when the joint backward will run)
                prec in ("bf16-s8", "f16") and self.joint_backward and self.tail_gmax)
            if u.dim() == 1 or u.stride(0) == 0:
                u = u.expand(rays.shape[0], c["Ni"])
            ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, weights=w, pred_hyp=pred, u=u, z_std=std)
            return loss, dict(img_loss=comps[0], carve=comps[1] if self.carving_active() else None,
                              img_loss0=comps[2], ret=ret, share=share, loss_report=loss.detach())
        if self.fused_loss and not c["joint"] and hyp_per_ray:
            # the whole loss (affine map of the hypotheses :954, both photometric terms, the carving term,
            # their sum :968-983, this rank's share) in one forward / one backward entry
            largs = (ret["rgb_map"], ret["rgb0"], target_s, ret["pred_hyp"], target_hyp, self.depth_scales,
                     self.depth_shifts, img_i, mask, c["mask_mode"] == "wild", self.carving_active(), c["w"], c["thr"],
                     share)
            if self._unit_loss_ready and torch.is_grad_enabled():
                # Trainer.step / GraphedTrainer: loss forward + backward as one launch pair (the scale / shift
                # rows of the bucket are written by it: begin_step() left them alone)
                self._unit_loss_ready = False
                loss, comps = ops.TrainLossUnitFn.apply(*largs, self._one)
            else:
                loss, comps = ops.TrainLossFn.apply(*largs)
            return loss, dict(img_loss=comps[0], carve=comps[1] if self.carving_active() else None,
                              img_loss0=comps[2], ret=ret, share=share, loss_report=loss.detach())
        if torch.is_tensor(img_i):
            # device index (GraphedTrainer keeps it in a static buffer): gather, no host read
            scale, shift = self.depth_scales.index_select(0, img_i.reshape(1)), \
                self.depth_shifts.index_select(0, img_i.reshape(1))
            target_h = target_hyp * scale.reshape(()) + shift.reshape(())
        else:
            target_h = target_hyp * self.depth_scales[img_i] + self.depth_shifts[img_i]      # :954
        mse_mask = mask if c["mask_mode"] == "wild" else None
        mse = (lambda a, b: H.img2mse(a, b)) if mse_mask is None else \
            (lambda a, b: H.img2mse_masked(a, b, mse_mask))
        img_loss = mse(ret["rgb_map"], target_s)                                              # :968
        loss = img_loss
        carve = None
        global_terms = None
        if self.carving_active():                                                             # :973
            carve = H.compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=c["joint"],
                                                 mask=mask, norm_p=c["norm_p"], threshold=c["thr"],
                                                 sharded=self.sharded, n_total=n_total)
            if c["joint"] and self.sharded:
                global_terms = c["w"] * carve      # already the loss of the WHOLE batch on every rank
            else:
                loss = loss + c["w"] * carve
        img_loss0 = mse(ret["rgb0"], target_s)                                                # :981
        loss = loss + img_loss0
        if share != 1.0:
            loss = loss * share
        report = loss.detach()
        if global_terms is not None:
            # a term that is already the whole batch's loss on every rank: full weight in the backward
            # (each rank's backward yields its shard's part), its share in the reported value so that
            # the ranks' reports still sum to the global loss
            loss = loss + global_terms
            report = report + share * global_terms.detach()
        return loss, dict(img_loss=img_loss, carve=carve, img_loss0=img_loss0, ret=ret, share=share,
                          loss_report=report)

    def reduce_grads(self):
        """The step's gradient exchange: RCCL sum-all-reduce of the bucket over the ranks' shards."""
        if not (self.sharded or self.force_allreduce):
            return
        if self.allreduce == "staged":
            # what the backward's hook has not sent yet (everything, when the backward was not a joint one)
            b, done = self.bucket, sorted(self._staged_done)
            pos, rest = 0, []
            for a0, n0 in done:
                if a0 > pos:
                    rest.append((pos, a0 - pos, None))
                pos = a0 + n0
            if pos < b.numel:
                rest.append((pos, b.numel - pos, None))
            works = self._staged_works + b.allreduce_grads_async(rest, force=self.force_allreduce)
            self._staged_works, self._staged_done = [], []
            b.wait_all(works)
        elif self.allreduce == "overlap" and self.coarse_stream is not None:
            b = self.bucket
            works = b.allreduce_grads_async(
                [(0, self.n_coarse, self.coarse_stream),                 # behind the coarse backward chain
                 (self.n_coarse, b.numel - self.n_coarse, None)],        # fine | scales | shifts
                force=self.force_allreduce)
            b.wait_all(works)
        else:
            self.bucket.allreduce_grads(force=self.force_allreduce)

    def _send_net_grads(self, net):
        """allreduce="staged": ``net``'s gradient is complete on the stream - start the all-reduce of its piece of the
        bucket (the last network's piece takes the scale / shift rows behind it along: the loss kernel wrote them
        long ago)."""
        b = self.bucket
        start = next((o for n, o in b._sinks if n is net), None)
        if start is None:
            return
        numel = ops.N_PARAM_FLOATS
        if start + numel == self.n_net:
            numel = b.numel - start
        self._staged_works += b.allreduce_grads_async([(start, numel, None)], force=self.force_allreduce)
        self._staged_done.append((start, numel))

    def begin(self):
        """Start of a step driven through forward_loss() / backward(): gradient sinks armed (FlatParams.begin_step)
        and - when the fused three-term loss will run in its unit-gradient form, which WRITES the scale / shift
        gradient rows - without the zero fill of those rows."""
        c = self.cfg
        # (needs both networks' gradient regions covered by sinks: what lies outside them is not zero-filled then)
        unit = self.fused_loss and not c["joint"] and len(self.bucket._sinks) == 2
        self.bucket.begin_step(zero_outside_sinks=not unit)
        self._unit_loss_ready = unit

    def backward(self, loss, defer_reduce=False):
        """loss.backward() (:985) with the two networks' MLP backwards joined into one launch sequence.
        ``defer_reduce`` (``step`` and the captured step pass True): where no gradient exchange follows, the joint
        launch leaves its partial rows unsummed and ``finish()`` - which MUST then follow - sums them inside the
        optimizer's launch; the bucket holds the step's gradient only after that.  False: the gradient is in the bucket
        when this returns."""
        self._pending_reduce = None
        if self.joint_backward:
            from .mlp_bwd import DeferredBackward
            exchange = self.sharded or self.force_allreduce
            staged = self.allreduce == "staged" and exchange
            self._staged_works, self._staged_done = [], []
            defer = defer_reduce and self.fused_finish and not exchange and len(self.bucket._sinks) == 2
            with DeferredBackward(after_net=self._send_net_grads if staged else None, defer_reduce=defer) as q:
                loss.backward(self._unit_grad(loss))
            if q.reduce is not None:
                desc, (n0, _) = q.reduce
                self._pending_reduce = desc.swap() if n0 is self.fine else desc
        else:
            loss.backward(self._unit_grad(loss))

    def step(self, rays, target_s, target_hyp, img_i=0, mask=None, n_total=None, **render_kw):
        """One optimisation step on this rank's shard; returns (this rank's term of the global loss,
        aux).  ``n_total``: rays of the whole batch when the shards are uneven."""
        self.begin()
        loss, aux = self.forward_loss(rays, target_s, target_hyp, img_i, mask, n_total, **render_kw)
        if self._unit_loss_ready:        # the loss took another form (cached-quantile hypotheses): zero the rows now
            self._unit_loss_ready = False
            self.flat_ss.grad.zero_()
        self.backward(loss, defer_reduce=True)                                                # :985
        self.bucket.end_backward()
        self.reduce_grads()
        lr = staircase_lr(self.cfg["lrate"], self.cfg["rate"], self.cfg["step"], self.it + 1)  # :988-991
        self.finish(lr_a=lr)
        self.it += 1
        return aux["loss_report"], aux

    def finish(self, lr_a=None, dev=False):
        """optimizer.step() (:993) and, while i < freeze_ss, optimizer_ss.step() (:996-997) as one launch - with the
        sum of a deferred backward's partial rows in front (ops.step_finish)."""
        opt_ss = self.opt_ss if self.scaleshift_active() else None
        desc, self._pending_reduce = self._pending_reduce, None
        if desc is None:
            return adam_step_pair(self.opt, opt_ss, lr_a=lr_a, dev=dev, ticked=dev)
        ops.step_finish(self.opt, opt_ss, 2, lr_a=lr_a, dev=dev, reduce=desc)
