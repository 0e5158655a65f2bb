"""Host-side I/O around the hot path: scene /
hypothesis loader, checkpoints in the reference's format, scene bounding box, and the
full-image evaluation loop.  File formats follow the reference exactly so its datasets and
pretrained checkpoints can be consumed unchanged:

  load_scene_scannet      data/load_scene.py:243-383 (read_files :16-26, gt depth :72-91)
  load_scene_processed    data/load_scene.py:386-532 (the in-the-wild scenes of run_scade_wild.py)
  scene_bbox              run_scade_scannet.py:1236-1244
  save/load_checkpoint    run_scade_scannet.py:411-420, :1004-1019
  render_images_with_metrics (PSNR + depth RMSE part)   run_scade_scannet.py:304-394
  write_images_with_metrics / MeanTracker               run_scade_scannet.py:396-409, train_utils/logging.py:5-34
  render_video (frame loop; the ffmpeg call only if present)   run_scade_scannet.py:236-264

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
    return _load_scene(basedir, cimle_dir, num_hypothesis, train_json, init_scales, scales_dir, gt_init, False)


def load_scene_processed(basedir, cimle_dir, num_hypothesis=20, train_json="transforms_train.json",
                         init_scales=False, scales_dir=None, gt_init=False):
    """The in-the-wild loader run_scade_wild.py:1261 uses (data/load_scene.py:386-532): the depth
    map of a frame is ``<depth_file_path stem>.png`` whatever extension the json names (:423), and
    there are no separate ground-truth depth maps (the 11th / 12th entries of the tuple are None)."""
    return _load_scene(basedir, cimle_dir, num_hypothesis, train_json, init_scales, scales_dir, gt_init, True)


SPLITS = ("train", "val", "test", "video")


def _frame_table(basedir, train_json, processed):
    """The scene's ``transforms_<split>.json`` files flattened into ONE table with a row per frame, in split order:
    split id, pose, intrinsics, and the two file names ('' for a frame that has none - the video poses).  The header
    values near / far / depth_scaling_factor are the training file's (data/load_scene.py:261-294)."""
    rows, header = [], {}
    for sid, split in enumerate(SPLITS):
        if not os.path.exists(os.path.join(basedir, f"transforms_{split}.json")):
            continue
        with open(os.path.join(basedir, train_json if split == "train" else f"transforms_{split}.json")) as fp:
            meta = json.load(fp)
        if split == "train":
            header = {k: float(meta[k]) for k in ("near", "far", "depth_scaling_factor")}
        for fr in meta["frames"]:
            rgb, dep = fr["file_path"], fr["depth_file_path"]
            if processed and (rgb or dep):
                dep = dep.split(".")[0] + ".png"                       # :423: always the .png beside the json's name
            rows.append((sid, fr["transform_matrix"], (fr["fx"], fr["fy"], fr["cx"], fr["cy"]), rgb, dep))
    return rows, header


def _load_scene(basedir, cimle_dir, num_hypothesis, train_json, init_scales, scales_dir, gt_init, processed):
    """Both loaders: the frame table first, then every array of the reference's return tuple assembled in one pass
    each - poses / intrinsics by one ``np.asarray`` over the table's columns, the split index lists from the split-id
    column, images / depth maps read frame by frame into arrays allocated once at their final size."""
    rows, header = _frame_table(basedir, train_json, processed)
    near, far, dsf = header["near"], header["far"], header["depth_scaling_factor"]
    sid = np.asarray([r[0] for r in rows], dtype=np.int64)
    poses = np.asarray([r[1] for r in rows], dtype=np.float32)
    intrinsics = np.asarray([r[2] for r in rows], dtype=np.float32)
    i_split = [np.flatnonzero(sid == k) for k in range(len(SPLITS))]
    # frames that name files (every split but the video poses), in table order = the order of the image arrays
    with_files = [r for r in rows if r[3] or r[4]]
    filenames = [r[3] for r in with_files]
    imgs = depths = valid_depths = None
    for n, r in enumerate(with_files):
        img, depth = read_files(basedir, r[3], r[4])
        depth = depth[..., None] if depth.ndim == 2 else depth
        if imgs is None:
            imgs = np.empty((len(with_files),) + img.shape, np.float32)
            depths = np.empty((len(with_files),) + depth.shape, np.float32)
            valid_depths = np.empty((len(with_files),) + depth.shape[:2], bool)
        imgs[n], depths[n], valid_depths[n] = img, depth / dsf, depth[:, :, 0] > 0.5
    Hh, Ww = imgs.shape[1:3]
    gt_depths, gt_valid_depths = (None, None) if processed else load_ground_truth_depth(basedir, filenames, (Hh, Ww), dsf)

    # K depth hypotheses per training view: train/leres_cimle/<cimle_dir>/<image id>_<j>.npy, clipped to [near, far]
    # (data/load_scene.py:319-348) -> [N_train, K, H, W, 1]
    stems = [filenames[idx].split("/")[-1].split(".")[0] for idx in i_split[0]]
    leres_dir = os.path.join(basedir, "train", "leres_cimle", cimle_dir)
    hyp = np.empty((len(stems), num_hypothesis, Hh, Ww, 1), np.float32)
    for v, stem in enumerate(stems):
        for j in range(num_hypothesis):
            hyp[v, j, :, :, 0] = np.load(os.path.join(leres_dir, f"{stem}_{j}.npy"))
    np.clip(hyp, near, far, out=hyp)
    ret = (imgs, depths, valid_depths, poses, Hh, Ww, intrinsics, near, far, i_split, gt_depths, gt_valid_depths, hyp)
    if init_scales:
        sdir = os.path.join(basedir, "train", "scale_shift_inits", scales_dir)
        ss = np.asarray([np.load(os.path.join(sdir, stem + ("_gtinit.npy" if gt_init else "_sfminit.npy")))
                         for stem in stems], dtype=np.float32)
        ret += (ss[:, 0], ss[:, 1])
    return ret


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


class MeanTracker:
    """Running means of named metrics (train_utils/logging.py:5-34: add / has / get / as_dict / print)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.mean_dict, self.total_weight = {}, 0

    def add(self, values, weight=1.):
        for k, v in values.items():
            self.mean_dict[k] = (self.mean_dict.get(k, 0) * self.total_weight + v) / (self.total_weight + weight)
        self.total_weight += weight

    def has(self, key):
        return key in self.mean_dict

    def get(self, key):
        return self.mean_dict[key]

    def as_dict(self):
        return self.mean_dict

    def print(self, f=None):
        for k, v in self.mean_dict.items():
            print("{}: {}".format(k, v), file=f) if f is not None else print("{}: {}".format(k, v))


@torch.no_grad()
def render_images_with_metrics(images, depths, valid_depths, poses, Hh, Ww, intrinsics, render_kwargs_test,
                               chunk=1024 * 16, count=None, indices=None, shard_group=None) -> Dict[str, object]:
    """The render + PSNR + depth-RMSE part of run_scade_scannet.py:304-394 (SSIM = skimage and
    LPIPS = AlexNet are third-party metrics and stay with the caller).  images [M,H,W,3],
    depths [M,H,W,1], valid_depths [M,H,W] are device tensors; returns per-image and mean metrics,
    the rendered rgb / depth maps, and under ``"images"`` / ``"mean_metrics"`` the result dict and
    tracker in the reference's layout (:380-394: channel-first CPU tensors, rgb clamped to [0,1],
    depths divided by ``far``) that ``write_images_with_metrics`` consumes.  ``shard_group``: every image's
    rays are split over the ranks of that torch.distributed group (True = default group) and the maps
    all-gathered, so each rank computes the same metrics from the same whole images."""
    idx = list(range(images.shape[0])) if indices is None else list(indices)
    if count is not None:
        idx = [int(i) for i in np.random.choice(idx, size=count, replace=False)]      # "take random images" (:312-314)
    far = float(render_kwargs_test.get("far", 1.0))
    out = {"psnr": [], "img_loss": [], "psnr0": [], "depth_rmse": [], "rgbs": [], "depths": []}
    res = {k: [] for k in ("rgbs", "target_rgbs", "depths", "target_depths", "target_valid_depths", "rgbs0", "depths0")}
    mean_metrics, mean_depth_metrics = MeanTracker(), MeanTracker()
    for n in idx:
        rgb, _, _, extras = R.render(Hh, Ww, intrinsics[n], chunk=chunk, c2w=poses[n], shard_group=shard_group,
                                     **render_kwargs_test)
        target = images[n]
        img_loss = H.img2mse(rgb, target)
        out["img_loss"].append(float(img_loss))
        out["psnr"].append(float(H.mse2psnr(img_loss)))
        metrics = {"img_loss": out["img_loss"][-1], "psnr": out["psnr"][-1]}
        if "rgb0" in extras:
            l0 = H.img2mse(extras["rgb0"], target)
            out["psnr0"].append(float(H.mse2psnr(l0)))
            metrics.update({"img_loss0": float(l0), "psnr0": out["psnr0"][-1]})
            res["rgbs0"].append(extras["rgb0"].clamp(0., 1.).permute(2, 0, 1).cpu())
            res["depths0"].append((extras["depth0"] / far).unsqueeze(0).cpu())
        v = valid_depths[n]
        if bool(v.any()):
            rmse = float(compute_rmse(extras["depth_map"][v], depths[n][:, :, 0][v]))
            if rmse == rmse:                                           # :347: NaN RMSEs are not tracked
                out["depth_rmse"].append(rmse)
                mean_depth_metrics.add({"depth_rmse": rmse})
        mean_metrics.add(metrics)
        out["rgbs"].append(rgb)
        out["depths"].append(extras["depth_map"])
        res["rgbs"].append(rgb.clamp(0., 1.).permute(2, 0, 1).cpu())
        res["target_rgbs"].append(target.permute(2, 0, 1).cpu())
        res["depths"].append((extras["depth_map"] / far).unsqueeze(0).cpu())
        res["target_depths"].append((depths[n][:, :, 0] / far).unsqueeze(0).cpu())
        res["target_valid_depths"].append(v.unsqueeze(0).cpu())
    out["mean"] = {k: float(np.mean(out[k])) for k in ("psnr", "img_loss", "psnr0", "depth_rmse") if out[k]}
    out["images"] = {k: torch.stack(v, 0) for k, v in res.items() if v}
    allm = MeanTracker()
    allm.add({**mean_metrics.as_dict(), **mean_depth_metrics.as_dict()})
    out["mean_metrics"] = allm
    return out


def write_images_with_metrics(images, mean_metrics, far, args=None, with_test_time_optimization=False,
                              result_dir=None):
    """run_scade_scannet.py:396-409: ``<n>_rgb.jpg`` (8 bit), ``<n>_d.png`` (16 bit, depth / far) and
    ``metrics.txt`` under ``ckpt_dir/expname/test_images_[with_optimization_]<scene_id>`` (``args`` as in
    the reference) or under an explicit ``result_dir``.  ``images`` / ``mean_metrics`` are the
    ``"images"`` / ``"mean_metrics"`` entries of ``render_images_with_metrics``.  PIL writes the files
    (the reference's cv2 is not part of this image; RGB order on disk is the same)."""
    from PIL import Image
    if result_dir is None:
        result_dir = os.path.join(args.ckpt_dir, args.expname, "test_images_" +
                                  ("with_optimization_" if with_test_time_optimization else "") + args.scene_id)
    os.makedirs(result_dir, exist_ok=True)
    rgbs = images["rgbs"].permute(0, 2, 3, 1).cpu().numpy()
    deps = images["depths"].permute(0, 2, 3, 1).cpu().numpy()
    for n, (rgb, depth) in enumerate(zip(rgbs, deps)):
        Image.fromarray(H.to8b(rgb)).save(os.path.join(result_dir, f"{n}_rgb.jpg"))
        Image.fromarray(H.to16b(depth[..., 0])).save(os.path.join(result_dir, f"{n}_d.png"))
    with open(os.path.join(result_dir, "metrics.txt"), "w") as f:
        mean_metrics.print(f)
    mean_metrics.print()
    return result_dir


# anchor colours of the two colormaps render_video paints with (cv2.COLORMAP_TURBO / COLORMAP_VIRIDIS in the
# reference), sampled at 0, 1/8, ..., 1 and interpolated linearly - a visualisation aid, not a metric
_TURBO = np.array([[48, 18, 59], [70, 107, 227], [40, 170, 235], [36, 227, 164], [143, 253, 74], [227, 228, 39],
                   [253, 157, 34], [223, 73, 11], [122, 4, 3]], np.float32)
_VIRIDIS = np.array([[68, 1, 84], [71, 45, 123], [59, 82, 139], [44, 114, 142], [33, 145, 140], [39, 173, 129],
                     [92, 200, 99], [170, 220, 50], [253, 231, 37]], np.float32)


def _colormap(x01, anchors):
    t = np.clip(np.asarray(x01, np.float32), 0., 1.) * (len(anchors) - 1)
    i = np.minimum(t.astype(np.int32), len(anchors) - 2)
    w = (t - i)[..., None]
    return (anchors[i] * (1 - w) + anchors[i + 1] * w).astype(np.uint8)


@torch.no_grad()
def render_video(poses, Hh, Ww, intrinsics, filename, render_kwargs_test, out_dir, chunk=1024 * 16, fps=25,
                 every=3, run_ffmpeg=True, write=True):
    """The frame loop of run_scade_scannet.py:236-264: every third pose rendered 16:9 with the centre
    third kept (``with_5_9``), frame = rgb | depth / far (turbo) | depth standard deviation (viridis),
    written as ``video_<filename>/<idx>.jpg``; ffmpeg assembles the mp4 when it is installed.
    Returns (frame directory, maximal depth seen).  With a sharded render (``shard_group`` in the kwargs) every
    rank calls this and takes part in each frame's render; only the rank with ``write=True`` touches the
    directory, the frames and ffmpeg."""
    import shutil
    import subprocess
    from PIL import Image
    video_dir = os.path.join(out_dir, "video_" + filename)
    if write:
        if os.path.exists(video_dir):
            shutil.rmtree(video_dir)
        os.makedirs(video_dir, exist_ok=True)
    depth_scale = float(render_kwargs_test["far"])
    max_depth = 0.0
    for img_idx in range(0, len(poses), every):
        rgb, _, _, extras = R.render(Hh, Ww, intrinsics[img_idx], chunk=chunk, c2w=poses[img_idx][:3, :4],
                                     with_5_9=True, **render_kwargs_test)
        max_depth = max(max_depth, float(extras["depth_map"].max()))
        if not write:
            continue
        frame = [H.to8b(rgb.cpu().numpy()),
                 _colormap((extras["depth_map"] / depth_scale).cpu().numpy(), _TURBO),
                 _colormap(depth_std_map(extras["z_vals"], extras["weights"], extras["depth_map"]).cpu().numpy(),
                           _VIRIDIS)]
        Image.fromarray(np.concatenate(frame, 1)).save(os.path.join(video_dir, f"{img_idx}.jpg"))
    if write and run_ffmpeg and shutil.which("ffmpeg"):
        subprocess.call(["ffmpeg", "-y", "-framerate", str(fps), "-i", os.path.join(video_dir, "%d.jpg"), "-c:v",
                         "libx264", "-profile:v", "high", "-crf", str(fps), os.path.join(out_dir, filename + ".mp4")])
    return video_dir, max_depth


def depth_std_map(z_vals, weights, depth_map):
    """Per-ray standard deviation of the rendered depth, the third panel of the reference's
    ``render_video`` frames (run_scade_scannet.py:257-258): sqrt(clamp(sum_i w_i (z_i - depth)^2,
    0, 1)).  A handful of elementwise torch ops on ``render``'s extras (visualisation, not on the
    kernel path; works on any device)."""
    var = ((z_vals - depth_map.unsqueeze(-1)).pow(2) * weights).sum(-1)
    return var.clamp(0.0, 1.0).sqrt()
