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

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import rendering as R
from . import run_nerf_helpers as H
from .optim import FusedAdam
from .parallel import FlatParams, shared_uniform, staircase_lr


def make_scade_nets(device, seed: Optional[int] = None):
    """coarse + fine NeRF in the SCADE configuration (create_nerf, :425-455)."""
    if seed is not None:
        torch.manual_seed(seed)
    mk = lambda: H.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3,
                        input_ch_cam=0, use_viewdirs=True).to(device)
    return mk(), mk()


class Trainer:
    def __init__(self, coarse, fine, bb_center, bb_scale, n_images=1, lrate=5e-4, scaleshift_lr=1e-7,
                 space_carving_weight=0.007, N_samples=64, N_importance=128, lrate_decay_rate=0.1,
                 lrate_decay_step=400000, freeze_ss=400000, norm_p=2, space_carving_threshold=0.0,
                 is_joint=False, warm_start_nerf=0, lindisp=False, raw_noise_std=0.0, precision="f32",
                 overlap_coarse=None):
        dev = next(coarse.parameters()).device
        self.coarse, self.fine = coarse, fine
        coarse.train_precision = fine.train_precision = precision      # "f32" (exact) | "f16x3"
        embed_fn, _ = H.get_embedder(9, 0)
        embeddirs_fn, _ = H.get_embedder(0, 0)
        self.query = R.make_network_query_fn(embed_fn, embeddirs_fn, bb_center.to(dev), bb_scale.to(dev))
        # DEPTH_SCALES / DEPTH_SHIFTS, one per training image (:878-888)
        self.depth_scales = torch.ones(n_images, 1, device=dev, requires_grad=True)
        self.depth_shifts = torch.zeros(n_images, 1, device=dev, requires_grad=True)
        self.flat = FlatParams(list(coarse.parameters()) + list(fine.parameters()))
        self.flat.attach_grad_sinks([coarse, fine])
        self.flat_ss = FlatParams([self.depth_scales, self.depth_shifts])
        self.opt = FusedAdam(self.flat, lr=lrate, betas=(0.9, 0.999))
        self.opt_ss = FusedAdam(self.flat_ss, lr=scaleshift_lr)
        self.cfg = dict(lrate=lrate, rate=lrate_decay_rate, step=lrate_decay_step, w=space_carving_weight,
                        Ns=N_samples, Ni=N_importance, freeze_ss=freeze_ss, norm_p=norm_p,
                        thr=space_carving_threshold, joint=is_joint, warm=warm_start_nerf,
                        lindisp=lindisp, noise=raw_noise_std)
        self.it = 0
        # coarse stage on a side stream: its backward chain then runs beside the fine one
        if overlap_coarse is None:
            overlap_coarse = os.environ.get("SCADE_OVERLAP_COARSE", "1") != "0"
        self.coarse_stream = torch.cuda.Stream(device=dev) if overlap_coarse and dev.type == "cuda" else None
        # rays are sharded over the ranks of the default process group (one process per GPU)
        self.sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.flat.broadcast_params(0)
        self.flat_ss.broadcast_params(0)

    def forward_loss(self, rays, target_s, target_hyp, img_i=0, mask=None, **render_kw):
        c = self.cfg
        target_h = target_hyp * self.depth_scales[img_i] + self.depth_shifts[img_i]          # :954
        if c["joint"] and self.sharded:
            # sample_pdf_joint draws ONE u[S] for the whole batch (helpers:452-453): rank 0's draw
            render_kw.setdefault("u_coarse", shared_uniform((c["Ni"],), rays.device))
            render_kw.setdefault("cached_u", shared_uniform((c["Ni"],), rays.device))
        elif not c["joint"] and not any(k in render_kw for k in ("t_rand", "u_coarse", "cached_u", "pytest")):
            # the step's three uniform draws (stratified jitter :564-579, the two sample_pdf draws
            # helpers:346-361) as ONE generator launch; independent streams either way
            n, ns, ni = rays.shape[0], c["Ns"], c["Ni"]
            d = torch.rand(n * (ns + 2 * ni), device=rays.device)
            render_kw["t_rand"] = d[:n * ns].view(n, ns)
            render_kw["u_coarse"] = d[n * ns:n * (ns + ni)].view(n, ni)
            render_kw["cached_u"] = d[n * (ns + ni):].view(n, ni)
        ret = R.render_rays(rays, True, self.coarse, self.query, c["Ns"], N_importance=c["Ni"],
                            network_fine=self.fine, perturb=1., raw_noise_std=c["noise"],
                            lindisp=c["lindisp"], is_joint=c["joint"], coarse_stream=self.coarse_stream,
                            **render_kw)
        mse = (lambda a, b: H.img2mse(a, b)) if mask is None else (lambda a, b: H.img2mse_masked(a, b, mask))
        img_loss = mse(ret["rgb_map"], target_s)                                              # :968
        loss = img_loss
        carve = None
        if c["w"] > 0. and self.it >= c["warm"]:                                              # :973
            carve = H.compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=c["joint"],
                                                 mask=mask, norm_p=c["norm_p"], threshold=c["thr"],
                                                 sharded=self.sharded)
            loss = loss + c["w"] * carve
        img_loss0 = mse(ret["rgb0"], target_s)                                                # :981
        loss = loss + img_loss0
        return loss, dict(img_loss=img_loss, carve=carve, img_loss0=img_loss0, ret=ret)

    def step(self, rays, target_s, target_hyp, img_i=0, mask=None, **render_kw):
        """One optimisation step on this rank's shard; returns the (local) loss tensor."""
        self.opt.zero_grad()
        self.opt_ss.zero_grad()
        loss, aux = self.forward_loss(rays, target_s, target_hyp, img_i, mask, **render_kw)
        loss.backward()                                                                       # :985
        gs = self.flat.allreduce_grads()
        gs_ss = self.flat_ss.allreduce_grads()
        lr = staircase_lr(self.cfg["lrate"], self.cfg["rate"], self.cfg["step"], self.it)     # :988-991
        self.opt.step(grad_scale=gs, lr=lr)                                                   # :993
        if self.it < self.cfg["freeze_ss"]:                                                   # :996-997
            self.opt_ss.step(grad_scale=gs_ss)
        self.it += 1
        return loss.detach(), aux
