"""Thin train / test driver over the HIP operators: the per-scene call sequence of the reference's
``train_nerf`` / ``run_nerf`` (run_scade_scannet.py:830-1089, :1207-1285; run_scade_wild.py likewise), one
process per GPU.  It is NOT a port of the reference's CLI (configargparse, tensorboard, lpips / skimage
metrics and ffmpeg stay with the caller): it ties the pieces this package owns -
scene loader, batch gather, ``Trainer.step``, checkpoints in the reference's format, the full-image eval
loop and the image writer - so that a scene directory goes in and checkpoints, test images and
``metrics.txt`` come out.

    python -m scade_amd.driver --data_dir <datasets/scannet> --scene_id scene0758_00 --cimle_dir dump_... \\
        --ckpt_dir checkpoints --expname demo --num_iterations 500000
    torchrun --nproc-per-node 8 -m scade_amd.driver ...        # rays of every batch sharded over the ranks

Per iteration i (counted from 1 like the reference, :899-900): pick a training image (numpy stream seeded
like :831), pick N_rand pixels without replacement (:786), gather ray rows / colours / K hypotheses / corner
mask for THOSE pixels only (the reference generates all H*W rays first), one train step (hypotheses * scale +
shift, render, three-term loss, backward, gradient all-reduce, both Adam updates, staircase learning rate,
scale/shift freeze).  By default the iteration is TWO host calls: one batch-gather launch that writes into the
static inputs of the graph-captured step (``scade_gather_batch``) and one ``graph.replay()``
(``GraphedTrainer.step_staged``); ``graph=False`` keeps the eager ``get_ray_batch`` + ``Trainer.step`` pair.  With a process group every rank takes its
contiguous share of the SAME N_rand pixels, so the global batch is the reference's batch.
"""
from __future__ import annotations

import argparse
import os
import time
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import ops, parallel, scene
from . import run_nerf_helpers as H
from .graphs import GraphedTrainer
from .train import Trainer, make_scade_nets


def render_kwargs_test(trainer: Trainer, near: float, far: float, precision: Optional[str] = None):
    """create_nerf's render_kwargs_test (:488-507): deterministic sampling, no noise."""
    coarse, fine = trainer.coarse, trainer.fine
    if precision is not None:
        coarse.inference_precision = fine.inference_precision = precision
    c = trainer.cfg
    return dict(network_query_fn=trainer.query, perturb=False, N_importance=c["Ni"], network_fine=fine,
                N_samples=c["Ns"], network_fn=coarse, use_viewdirs=True, raw_noise_std=0., lindisp=c["lindisp"],
                near=near, far=far)


def train_scene(data, out_dir: str, expname: str = "scade", scene_id: str = "scene", num_iterations: int = 500000,
                N_rand: int = 1024,
                i_weights: int = 100000, i_print: int = 1000, mask_corners: bool = False, mask_edges: bool = False,
                wild: bool = False, scale_init: float = 1.0, shift_init: float = 0.0, scales_init=None,
                shifts_init=None, seed: int = 0, precision: str = "f32", eval_precision: Optional[str] = None,
                test_chunk: int = 1024 * 16, no_reload: bool = False, log=print, pixel_sampler: str = "device",
                i_img: int = 0, n_val_images: int = 8, graph: Optional[bool] = None, loop_warmup: int = 0,
                tail_losses: int = 0, **trainer_kw):
    """``data`` = the tuple of scene.load_scene_scannet / load_scene_processed.  Returns a dict with the
    trainer, the loss trace and (rank 0) the test metrics.  ``trainer_kw`` goes to ``Trainer`` (lrate,
    scaleshift_lr, space_carving_weight, is_joint, warm_start_nerf, freeze_ss, allreduce, ...).
    ``pixel_sampler``: "device" takes the N_rand pixels of a step as the next slice of a torch.randperm of the
    H*W pixels drawn on the GPU from a generator seeded identically on every rank - every batch is a uniform
    N_rand-subset without replacement like the reference's, and one 150 us permutation serves H*W / N_rand
    steps; "numpy" is the reference's
    ``np.random.choice(H*W, N_rand, replace=False)`` (:786) - 4 ms of host time per step at 468 x 624, which
    is more than a whole bf16 train step.  ``i_img`` > 0: every i_img iterations the first ``n_val_images``
    validation views (the test views when the scene has no validation split, :854-857) are rendered and their
    mean metrics logged and kept in the result (:1036-1045).
    ``graph``: the iteration as ONE batch-gather launch + ONE HIP-graph replay (``GraphedTrainer.step_staged``): the
    launch gathers the N_rand pixels' ray rows / colours / K hypotheses / mask of the picked view from the resident
    training set straight into the captured step's static buffers, stores the view's index and advances the
    optimizers' device scalars.  ``None`` = on, except where the step cannot be captured (a process group whose
    backend is not RCCL; the joint loss over ranks, whose shared draw is a host-side broadcast).  ``False`` = the
    eager ``Trainer.step`` on a batch from ``get_ray_batch`` (same pixels, same draws, same arithmetic: a test holds
    the two loops' parameters equal).  ``loop_warmup``: iterations left out of ``ms_per_iteration`` (graph capture,
    allocator warm-up, clock ramp) - the figure is then the steady-state rate.  ``tail_losses`` > 0: the loss of each
    of the last ``tail_losses`` iterations is kept (one device copy per iteration, no host read) and their mean is
    returned as ``tail_loss_mean``; those iterations are left out of ``ms_per_iteration``."""
    imgs, depths, valid, poses, Hh, Ww, intr, near, far, i_split, gt_d, gt_v, hyps = data[:13]
    if len(data) >= 15 and scales_init is None:
        scales_init, shifts_init = data[13], data[14]
    dev = torch.device("cuda", torch.cuda.current_device())
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    i_train, i_val, i_test = i_split[0], i_split[1], i_split[2]
    if len(i_test) == 0:
        raise ValueError("train_scene: the scene has no test split (:851-853)")
    if gt_d is not None:                       # ground-truth depth for validation / test when there is any (:843-847)
        depths, valid = np.array(depths, copy=True), np.array(valid, copy=True)   # the caller's arrays stay as loaded
        for ix in (i_test, i_val):
            depths[ix], valid[ix] = gt_d[ix], gt_v[ix]
    def to(a, dt=torch.float32):
        """-> device tensor (numpy arrays from the loaders, or tensors a caller already holds on the device)"""
        if torch.is_tensor(a):
            return a.to(device=dev, dtype=dt)
        return torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
    # the whole training set is resident (288 GB of HBM: a ScanNet scene is a few hundred MB)
    t_img, t_pose, t_intr = to(imgs[i_train]).contiguous(), to(poses[i_train]).contiguous(), to(intr[i_train]).contiguous()
    t_hyp = to(hyps).contiguous()                                           # [N_train, K, H, W, 1]
    n_train = t_img.shape[0]

    np.random.seed(seed)                                                    # :831-833
    coarse, fine = make_scade_nets(dev, seed=seed)
    bbc, bbs = scene.scene_bbox(Hh, Ww, intr, poses, i_train, far, dev)      # :1236-1244
    ckpt = scene.load_checkpoint(out_dir, expname, no_reload=no_reload)
    start = scene.restore(coarse, fine, ckpt) if ckpt is not None else 0
    tr = Trainer(coarse, fine, bbc, bbs, n_images=n_train, precision=precision, start_iter=start,
                 mask_mode="wild" if wild else "scannet", **trainer_kw)
    with torch.no_grad():
        if ckpt is not None and "depth_scales" in ckpt:                     # :931-933
            tr.depth_scales.copy_(ckpt["depth_scales"].to(dev)[:n_train].reshape(n_train, 1))
            tr.depth_shifts.copy_(ckpt["depth_shifts"].to(dev)[:n_train].reshape(n_train, 1))
        elif scales_init is not None:
            tr.depth_scales.copy_(to(scales_init).reshape(n_train, 1))
            tr.depth_shifts.copy_(to(shifts_init).reshape(n_train, 1))
        else:
            tr.depth_scales.fill_(scale_init)
            tr.depth_shifts.fill_(shift_init)
    parallel.seed_rank_streams(seed + 1)       # identical weights everywhere, per-rank jitter / u streams
    all_coords = torch.stack(torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij"), -1).reshape(-1, 2)
    if pixel_sampler not in ("device", "numpy"):
        raise ValueError('train_scene: pixel_sampler must be "device" or "numpy"')
    all_coords_dev = all_coords.to(dev)
    g_pix = torch.Generator(device=dev).manual_seed(seed + 12345)         # the SAME stream on every rank
    perm, cursor = None, 0

    i_eval = i_val if len(i_val) > 0 else i_test
    val = None
    if i_img > 0:
        ix = i_eval[:n_val_images]
        val = (to(imgs[ix]), to(depths[ix]), to(valid[ix], torch.bool), to(poses[ix]), to(intr[ix]))
    val_trace = []
    a, b = parallel.shard_range(N_rand, rank, world)                        # this rank's share of every batch
    if graph is None:
        graph = not (world > 1 and (dist.get_backend() != "nccl" or trainer_kw.get("is_joint", False)))
    gt = gather = None
    if graph:
        masked = bool(mask_corners or mask_edges)
        gt = GraphedTrainer(tr, b - a, t_hyp.shape[1], n_total=N_rand, with_mask=masked)
        gather = ops.ResidentBatchGather(Hh, Ww, t_img, t_hyp, t_pose, t_intr, near, far, gt.rays, gt.tgt, gt.hyp,
                                         gt.mask, corner_px=20 if mask_corners else 0,
                                         edge_px=10 if (mask_edges and not mask_corners) else 0,      # (an elif: run_scade_wild.py:818)
                                         scalar_dst=gt.img_i, tick_states=(tr.opt, tr.opt_ss), points=gt.coarse_pre,
                                         packs=gt.packs if gt.coarse_pre is not None else None)
        if gt.packs is not None and gt.coarse_pre is not None:
            gt.opening_packs = True       # (this loop's opening launch is the gather: it packs)
            gt.packs.prepare()            # the blobs exist before the first gather launch re-packs them
    trace, t0, t_aux, i0 = [], time.time(), 0.0, start     # t_aux: validation renders + checkpoint writes, not loop time
    tail_losses = max(0, min(int(tail_losses), num_iterations - start - loop_warmup - 1))
    tail_buf = torch.zeros(max(1, tail_losses), device=dev)
    i_tail, t_end = num_iterations - tail_losses, None     # the timed span ends behind iteration i_tail
    for i in range(start + 1, num_iterations + 1):
        if loop_warmup > 0 and i == start + 1 + loop_warmup:
            torch.cuda.synchronize()
            t0, t_aux, i0 = time.time(), 0.0, i - 1
        img_i = int(np.random.choice(n_train))                             # :946 (same stream on every rank)
        if pixel_sampler == "numpy":
            sel = np.random.choice(Hh * Ww, size=[N_rand], replace=False)  # :786 / helpers:279-283
            pix, off = torch.from_numpy(sel[a:b]).to(dev), 0
        else:
            if perm is None or cursor + N_rand > Hh * Ww:
                perm, cursor = torch.randperm(Hh * Ww, generator=g_pix, device=dev), 0
            pix, off = perm, cursor + a
            cursor += N_rand
        if graph:
            gather(pix, off, img_i, tick_second=tr.scaleshift_active())
            loss = gt.step_staged()
        else:
            coords = all_coords_dev[pix[off:off + b - a]]
            rays, target_s, target_h, mask = H.get_ray_batch(
                Hh, Ww, t_intr[img_i], t_pose[img_i], coords, near, far, image=t_img[img_i], hypotheses=t_hyp[img_i],
                mask_corners=mask_corners, mask_edges=mask_edges)
            loss, aux = tr.step(rays, target_s, target_h, img_i=img_i, mask=mask, n_total=N_rand)
        if tail_losses and i >= i_tail:
            if i == i_tail:
                torch.cuda.synchronize()
                t_end = time.time()
            else:
                tail_buf[i - i_tail - 1].copy_(loss.detach().reshape(()))
        if i % i_print == 0 or i == num_iterations:
            lv = float(loss)
            trace.append((i, lv))
            if rank == 0:
                img_loss = gt.terms[0] if graph else aux["img_loss"]
                log(f"[TRAIN] iter {i}  loss (this rank's term) {lv:.6f}  psnr {float(H.mse2psnr(img_loss)):.2f}"
                    f"  {(time.time() - t0 - t_aux) / max(1, i - i0) * 1e3:.2f} ms/it")
        if val is not None and i % i_img == 0:                              # :1036-1045
            torch.cuda.synchronize()
            ta = time.time()
            m = scene.render_images_with_metrics(val[0], val[1], val[2], val[3], Hh, Ww, val[4],
                                                 render_kwargs_test(tr, near, far, eval_precision), chunk=test_chunk,
                                                 shard_group=True if world > 1 else None)["mean"]
            val_trace.append((i, m))
            if rank == 0:
                log(f"[VAL] iter {i}  {m}")
            torch.cuda.synchronize()
            t_aux += time.time() - ta
        if i % i_weights == 0 and rank == 0:                                # :1004-1021, reference key names
            torch.cuda.synchronize()
            ta = time.time()
            path = os.path.join(out_dir, expname, "{:06d}.tar".format(i))
            scene.save_checkpoint(path, i, coarse, fine, tr.depth_shifts, tr.depth_scales)
            log("Saved checkpoints at " + path)
            t_aux += time.time() - ta

    torch.cuda.synchronize()
    i_last = i_tail if t_end is not None else num_iterations
    loop_ms = ((t_end if t_end is not None else time.time()) - t0 - t_aux) * 1e3 / max(1, i_last - i0)

    # ---- test at the last iteration (:1071-1086): every test image, metrics, images on disk -------------
    kw = render_kwargs_test(tr, near, far, eval_precision)
    group = True if world > 1 else None
    res = scene.render_images_with_metrics(to(imgs[i_test]), to(depths[i_test]), to(valid[i_test], torch.bool),
                                           to(poses[i_test]), Hh, Ww, to(intr[i_test]), kw,
                                           chunk=test_chunk, shard_group=group)
    out = {"trainer": tr, "trace": trace, "test": res["mean"], "iterations": num_iterations,
           "ms_per_iteration": loop_ms, "val": val_trace, "graphed": bool(graph),
           "iterations_timed": i_last - i0}
    if tail_losses:
        out["tail_loss_mean"], out["tail_losses"] = float(tail_buf[:tail_losses].mean()), tail_losses
    if rank == 0:
        args = SimpleNamespace(ckpt_dir=out_dir, expname=expname, scene_id=scene_id)
        scene.write_images_with_metrics(res["images"], res["mean_metrics"], far, args)
        log(f"[TEST] {res['mean']}")
    return out


def test_scene(data, ckpt_dir: str, expname: str, scene_id: str = "scene", task: str = "test",
               precision: Optional[str] = None, test_chunk: int = 1024 * 16, log=print):
    """``--task test`` / ``--task video`` of the reference's run_nerf (:1262-1282): the latest checkpoint under
    ckpt_dir/expname (the reference's own files load: ``module.``-prefixed keys), then every test view rendered,
    scored and written (``test``), or the video poses rendered as frames (``video``).  With a process group the
    rays of every image are sharded over the ranks."""
    imgs, depths, valid, poses, Hh, Ww, intr, near, far, i_split, gt_d, gt_v, _ = data[:13]
    dev = torch.device("cuda", torch.cuda.current_device())
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    to = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
    coarse, fine = make_scade_nets(dev, seed=0)
    ckpt = scene.load_checkpoint(ckpt_dir, expname)
    if ckpt is None:
        raise FileNotFoundError(f"no '*000.tar' checkpoint under {os.path.join(ckpt_dir, expname)}")
    step = scene.restore(coarse, fine, ckpt)
    bbc, bbs = scene.scene_bbox(Hh, Ww, intr, poses, i_split[0], far, dev)
    tr = Trainer(coarse, fine, bbc, bbs, n_images=1, start_iter=step)      # query fn + sample counts; never stepped
    kw = render_kwargs_test(tr, near, far, precision)
    group = True if world > 1 else None
    if task == "video":
        i_video = i_split[3]
        kwv = dict(kw, shard_group=group, shard_keys=None) if group else kw
        out_dir = os.path.join(ckpt_dir, expname)
        return scene.render_video(to(poses[i_video]), Hh, Ww, to(intr[i_video]), "0", kwv, out_dir, chunk=test_chunk,
                                  run_ffmpeg=rank == 0, write=rank == 0)
    i_test = i_split[2]
    d, v = (depths, valid) if gt_d is None else (gt_d, gt_v)               # :1268-1273
    with torch.no_grad():
        res = scene.render_images_with_metrics(to(imgs[i_test]), to(d[i_test]), to(v[i_test], torch.bool),
                                               to(poses[i_test]), Hh, Ww, to(intr[i_test]), kw, chunk=test_chunk,
                                               shard_group=group)
    if rank == 0:
        args = SimpleNamespace(ckpt_dir=ckpt_dir, expname=expname, scene_id=scene_id)
        scene.write_images_with_metrics(res["images"], res["mean_metrics"], far, args)
        log(f"[TEST] checkpoint step {step}: {res['mean']}")
    return res


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("--task", default="train", choices=["train", "test", "video"])
    p.add_argument("--data_dir", required=True)
    p.add_argument("--scene_id", required=True)
    p.add_argument("--cimle_dir", required=True)
    p.add_argument("--dataset", default="scannet", choices=["scannet", "processed"],
                   help="scannet: load_scene_scannet; processed: the in-the-wild loader (run_scade_wild.py)")
    p.add_argument("--ckpt_dir", default="checkpoints")
    p.add_argument("--expname", default="scade")
    p.add_argument("--num_hypothesis", type=int, default=20)
    p.add_argument("--N_rand", type=int, default=1024)
    p.add_argument("--num_iterations", type=int, default=500000)
    p.add_argument("--i_weights", type=int, default=100000)
    p.add_argument("--i_print", type=int, default=1000)
    p.add_argument("--i_img", type=int, default=20000, help="validation render every i_img iterations (0: never)")
    p.add_argument("--lrate", type=float, default=5e-4)
    p.add_argument("--scaleshift_lr", type=float, default=None, help="default 1e-7 (scannet) / 1e-5 (processed)")
    p.add_argument("--space_carving_weight", type=float, default=0.007)
    p.add_argument("--warm_start_nerf", type=int, default=0)
    p.add_argument("--freeze_ss", type=int, default=400000)
    p.add_argument("--is_joint", action="store_true")
    p.add_argument("--mask_corners", action="store_true")
    p.add_argument("--mask_edges", action="store_true", help="10-px border mask of run_scade_wild.py (:1220; off by default there too)")
    p.add_argument("--precision", default="f32", choices=["f32", "f16x3", "bf16", "bf16-s8", "f16"])
    p.add_argument("--eval_precision", default=None, choices=[None, "f32", "f16x3", "bf16", "f16"])
    p.add_argument("--no_reload", action="store_true")
    p.add_argument("--no_graph", action="store_true", help="eager Trainer.step per iteration instead of the HIP-graph replay")
    a = p.parse_args(argv)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    wild = a.dataset == "processed"
    load = scene.load_scene_processed if wild else scene.load_scene_scannet
    data = load(os.path.join(a.data_dir, a.scene_id), a.cimle_dir, a.num_hypothesis)
    if a.task != "train":
        test_scene(data, a.ckpt_dir, a.expname, a.scene_id, a.task, a.eval_precision)
        return
    train_scene(data, a.ckpt_dir, a.expname, a.scene_id, a.num_iterations, a.N_rand, a.i_weights, a.i_print,
                mask_corners=a.mask_corners, mask_edges=a.mask_edges, wild=wild, precision=a.precision,
                eval_precision=a.eval_precision, no_reload=a.no_reload, i_img=a.i_img, lrate=a.lrate,
                scaleshift_lr=a.scaleshift_lr if a.scaleshift_lr is not None else (1e-5 if wild else 1e-7),
                space_carving_weight=a.space_carving_weight, warm_start_nerf=a.warm_start_nerf,
                freeze_ss=a.freeze_ss, is_joint=a.is_joint, graph=False if a.no_graph else None)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
