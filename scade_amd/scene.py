"""Host-side I/O around the hot path: scene /
hypothesis loader, checkpoints in the reference's format, scene bounding box, and the
full-image evaluation loop.  File formats follow the reference exactly so its datasets and
pretrained checkpoints can be consumed unchanged:

  load_scene_scannet      data/load_scene.py:243-383 (read_files :16-26, gt depth :72-91)
  scene_bbox              run_scade_scannet.py:1236-1244
  save/load_checkpoint    run_scade_scannet.py:411-420, :1004-1019
  render_images_with_metrics (PSNR + depth RMSE part)   run_scade_scannet.py:304-394

Images are read with PIL (cv2 / imageio are not part of this image).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import rendering as R
from . import run_nerf_helpers as H


def read_files(basedir, rgb_file, depth_file):
    """RGB(A) in [0,1] float32 and raw depth as float64 (data/load_scene.py:16-26)."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(basedir, rgb_file)))
    if img.ndim == 2:
        img = np.stack([img] * 3, -1)
    img = (img / 255.).astype(np.float32)
    depth = np.asarray(Image.open(os.path.join(basedir, depth_file))).astype(np.float64)
    return img, depth


def load_ground_truth_depth(basedir, train_filenames, image_size, depth_scaling_factor):
    """data/load_scene.py:72-91."""
    from PIL import Image
    Hh, Ww = image_size
    gt_depths, gt_valid = [], []
    for filename in train_filenames:
        filename = filename.replace("rgb", "target_depth").replace(".jpg", ".png")
        f = os.path.join(basedir, filename)
        if os.path.exists(f):
            d = np.asarray(Image.open(f)).astype(np.float64)
            v = d > 0.5
            d = (d / depth_scaling_factor).astype(np.float32)
        else:
            d = np.zeros((Hh, Ww))
            v = np.full_like(d, False)
        gt_depths.append(np.expand_dims(d, -1))
        gt_valid.append(v)
    return np.stack(gt_depths, 0), np.stack(gt_valid, 0)


def load_scene_scannet(basedir, cimle_dir, num_hypothesis=20, train_json="transforms_train.json",
                       init_scales=False, scales_dir=None, gt_init=False):
    """Same return tuple as the reference (data/load_scene.py:243-383)."""
    splits = ['train', 'val', 'test', 'video']
    all_imgs, all_depths, all_valid, all_poses, all_intr = [], [], [], [], []
    counts, filenames = [0], []
    near = far = depth_scaling_factor = None
    Hh = Ww = None
    for s in splits:
        if not os.path.exists(os.path.join(basedir, f'transforms_{s}.json')):
            counts.append(counts[-1])
            continue
        jf = os.path.join(basedir, train_json if s == "train" else f'transforms_{s}.json')
        with open(jf) as fp:
            meta = json.load(fp)
        if 'train' in s:
            near, far = float(meta['near']), float(meta['far'])
            depth_scaling_factor = float(meta['depth_scaling_factor'])
        imgs, depths, valids, poses, intr = [], [], [], [], []
        for frame in meta['frames']:
            if len(frame['file_path']) != 0 or len(frame['depth_file_path']) != 0:
                img, depth = read_files(basedir, frame['file_path'], frame['depth_file_path'])
                if depth.ndim == 2:
                    depth = np.expand_dims(depth, -1)
                valids.append(depth[:, :, 0] > 0.5)
                depths.append((depth / depth_scaling_factor).astype(np.float32))
                filenames.append(frame['file_path'])
                imgs.append(img)
                Hh, Ww = img.shape[:2]
            poses.append(np.array(frame['transform_matrix']))
            intr.append(np.array((frame['fx'], frame['fy'], frame['cx'], frame['cy'])))
        counts.append(counts[-1] + len(poses))
        if imgs:
            all_imgs.append(np.array(imgs)); all_depths.append(np.array(depths)); all_valid.append(np.array(valids))
        all_poses.append(np.array(poses).astype(np.float32))
        all_intr.append(np.array(intr).astype(np.float32))
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(len(splits))]
    imgs = np.concatenate(all_imgs, 0)
    depths = np.concatenate(all_depths, 0)
    valid_depths = np.concatenate(all_valid, 0)
    poses = np.concatenate(all_poses, 0)
    intrinsics = np.concatenate(all_intr, 0)
    gt_depths, gt_valid_depths = load_ground_truth_depth(basedir, filenames, (Hh, Ww), depth_scaling_factor)

    leres_dir = os.path.join(basedir, "train", "leres_cimle", cimle_dir)
    hyps = []
    for idx in i_split[0]:
        img_id = filenames[idx].split("/")[-1].split(".")[0]
        cur = [np.expand_dims(np.load(os.path.join(leres_dir, f"{img_id}_{j}.npy")).astype(np.float32), -1)
               for j in range(num_hypothesis)]
        hyps.append(np.array(cur))
    all_depth_hypothesis = np.clip(np.array(hyps), near, far)          # [N_train, K, H, W, 1]
    ret = [imgs, depths, valid_depths, poses, Hh, Ww, intrinsics, near, far, i_split, gt_depths,
           gt_valid_depths, all_depth_hypothesis]
    if init_scales:
        sdir = os.path.join(basedir, "train", "scale_shift_inits", scales_dir)
        sc, sh = [], []
        for idx in i_split[0]:
            img_id = filenames[idx].split("/")[-1].split(".")[0]
            ss = np.load(os.path.join(sdir, img_id + ("_gtinit.npy" if gt_init else "_sfminit.npy"))).astype(np.float32)
            sc.append(ss[0]); sh.append(ss[1])
        ret += [np.array(sc), np.array(sh)]
    return tuple(ret)


def scene_bbox(Hh, Ww, intrinsics, poses, i_train, far, device):
    """bb_center / bb_scale from the far points of every training ray (:1236-1244)."""
    max_xyz = torch.full((3,), -1e6, device=device)
    min_xyz = torch.full((3,), 1e6, device=device)
    for idx in i_train:
        rays_o, rays_d = H.get_rays(Hh, Ww, torch.as_tensor(intrinsics[idx], dtype=torch.float32, device=device),
                                    torch.as_tensor(poses[idx], dtype=torch.float32, device=device))
        pts = (rays_o + rays_d * far).view(-1, 3)
        max_xyz = torch.max(pts.amax(0), max_xyz)
        min_xyz = torch.min(pts.amin(0), min_xyz)
    return (max_xyz + min_xyz) / 2., 2. / (max_xyz - min_xyz).max()


# ---------------------------------------------------------------------------
# checkpoints (reference format: DataParallel 'module.' prefixed state dicts)
# ---------------------------------------------------------------------------

def _with_module_prefix(sd):
    return {("module." + k): v.detach().cpu() for k, v in sd.items()}


def save_checkpoint(path, global_step, coarse, fine, depth_shifts, depth_scales, optimizer_state=None):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({'global_step': global_step,
                'network_fn_state_dict': _with_module_prefix(coarse.state_dict()),
                'network_fine_state_dict': _with_module_prefix(fine.state_dict()),
                'optimizer_state_dict': optimizer_state if optimizer_state is not None else {},
                'depth_shifts': depth_shifts.detach().cpu(), 'depth_scales': depth_scales.detach().cpu()}, path)


def load_checkpoint(ckpt_dir, expname, no_reload=False, map_location="cpu"):
    """Latest '*000.tar' under ckpt_dir/expname (run_scade_scannet.py:411-420)."""
    path = os.path.join(ckpt_dir, expname)
    if not os.path.isdir(path):
        return None
    ckpts = [os.path.join(path, f) for f in sorted(os.listdir(path)) if '000.tar' in f]
    if not ckpts or no_reload:
        return None
    return torch.load(ckpts[-1], map_location=map_location, weights_only=False)


def restore(coarse, fine, ckpt):
    """Weights only, like the reference (:477-486; the optimizer is not restored, :480)."""
    coarse.load_reference_state_dict(ckpt['network_fn_state_dict'])
    if fine is not None and 'network_fine_state_dict' in ckpt:
        fine.load_reference_state_dict(ckpt['network_fine_state_dict'])
    return ckpt.get('global_step', 0)


# ---------------------------------------------------------------------------
# evaluation (PSNR / depth RMSE of full-image renders)
# ---------------------------------------------------------------------------

def compute_rmse(prediction, target):
    """metric/rmse.py:3."""
    return torch.sqrt((prediction - target).pow(2).mean())


@torch.no_grad()
def render_images_with_metrics(images, depths, valid_depths, poses, Hh, Ww, intrinsics, render_kwargs_test,
                               chunk=1024 * 16, count=None, indices=None) -> Dict[str, object]:
    """The render + PSNR + depth-RMSE part of run_scade_scannet.py:304-394 (SSIM = skimage and
    LPIPS = AlexNet are third-party metrics and stay with the caller).  images [M,H,W,3],
    depths [M,H,W,1], valid_depths [M,H,W] are device tensors; returns per-image and mean metrics
    plus the rendered rgb / depth maps."""
    idx = list(range(images.shape[0])) if indices is None else list(indices)
    if count is not None:
        idx = idx[:count]
    out = {"psnr": [], "img_loss": [], "psnr0": [], "depth_rmse": [], "rgbs": [], "depths": []}
    for n in idx:
        rgb, _, _, extras = R.render(Hh, Ww, intrinsics[n], chunk=chunk, c2w=poses[n], **render_kwargs_test)
        target = images[n]
        img_loss = H.img2mse(rgb, target)
        out["img_loss"].append(float(img_loss))
        out["psnr"].append(float(H.mse2psnr(img_loss)))
        if "rgb0" in extras:
            out["psnr0"].append(float(H.mse2psnr(H.img2mse(extras["rgb0"], target))))
        v = valid_depths[n]
        if bool(v.any()):
            out["depth_rmse"].append(float(compute_rmse(extras["depth_map"][v], depths[n][:, :, 0][v])))
        out["rgbs"].append(rgb)
        out["depths"].append(extras["depth_map"])
    out["mean"] = {k: float(np.mean(out[k])) for k in ("psnr", "img_loss", "psnr0", "depth_rmse") if out[k]}
    return out


def depth_std_map(z_vals, weights, depth_map):
    """Per-ray standard deviation of the rendered depth, the third panel of the reference's
    ``render_video`` frames (run_scade_scannet.py:257-258): sqrt(clamp(sum_i w_i (z_i - depth)^2,
    0, 1)).  A handful of elementwise torch ops on ``render``'s extras (visualisation, not on the
    kernel path; works on any device)."""
    var = ((z_vals - depth_map.unsqueeze(-1)).pow(2) * weights).sum(-1)
    return var.clamp(0.0, 1.0).sqrt()
