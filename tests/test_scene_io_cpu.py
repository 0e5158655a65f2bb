"""CPU: the scene / checkpoint I/O (host logic) on a tiny synthetic
scene written to disk in the reference's file formats."""
import json
import os

import numpy as np
import pytest
import torch


def write_scene(root, Hh=12, Ww=16, n_train=2, n_test=1, K=3):
    from PIL import Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "train", "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "train", "depth"), exist_ok=True)
    os.makedirs(os.path.join(root, "test", "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "test", "depth"), exist_ok=True)
    os.makedirs(os.path.join(root, "train", "leres_cimle", "dump"), exist_ok=True)

    def frames(split, n):
        fr = []
        for i in range(n):
            rgb = (rng.rand(Hh, Ww, 3) * 255).astype(np.uint8)
            dep = (rng.rand(Hh, Ww) * 3000 + 500).astype(np.uint16)
            dep[0, 0] = 0                                              # invalid depth pixel
            Image.fromarray(rgb).save(os.path.join(root, split, "rgb", f"{i}.png"))
            Image.fromarray(dep).save(os.path.join(root, split, "depth", f"{i}.png"))
            pose = np.eye(4)
            pose[:3, 3] = [0.1 * i, 0.0, 0.2]
            fr.append({"file_path": f"{split}/rgb/{i}.png", "depth_file_path": f"{split}/depth/{i}.png",
                       "transform_matrix": pose.tolist(), "fx": 14.0, "fy": 14.0, "cx": Ww / 2, "cy": Hh / 2})
        return fr
    for split, n in (("train", n_train), ("test", n_test)):
        meta = {"near": 0.1, "far": 5.0, "depth_scaling_factor": 1000.0, "frames": frames(split, n)}
        json.dump(meta, open(os.path.join(root, f"transforms_{split}.json"), "w"))
    for i in range(n_train):
        for j in range(K):
            np.save(os.path.join(root, "train", "leres_cimle", "dump", f"{i}_{j}.npy"),
                    (rng.rand(Hh, Ww) * 8).astype(np.float32))        # some beyond far -> clipped
    return Hh, Ww


def test_load_scene_scannet(tmp_path):
    from scade_amd.scene import load_scene_scannet
    Hh, Ww = write_scene(str(tmp_path))
    (imgs, depths, valid, poses, H2, W2, intr, near, far, i_split, gt_d, gt_v, hyp) = \
        load_scene_scannet(str(tmp_path), "dump", num_hypothesis=3)
    assert (H2, W2) == (Hh, Ww) and (near, far) == (0.1, 5.0)
    assert imgs.shape == (3, Hh, Ww, 3) and imgs.dtype == np.float32 and imgs.max() <= 1.0
    assert depths.shape == (3, Hh, Ww, 1) and not valid[0, 0, 0] and valid[0, 1, 1]
    assert poses.shape == (3, 4, 4) and intr.shape == (3, 4)
    assert [len(s) for s in i_split] == [2, 0, 1, 0]
    assert hyp.shape == (2, 3, Hh, Ww, 1) and hyp.min() >= near and hyp.max() <= far
    assert gt_d.shape == (3, Hh, Ww, 1) and not gt_v.any()


def test_load_scene_processed(tmp_path):
    """data/load_scene.py:386-532: depth read from <depth_file_path stem>.png, no ground-truth depth maps."""
    from scade_amd.scene import load_scene_processed, load_scene_scannet
    Hh, Ww = write_scene(str(tmp_path))
    for split in ("train", "test"):                                # the json names another extension (:423)
        jf = os.path.join(str(tmp_path), f"transforms_{split}.json")
        meta = json.load(open(jf))
        for fr in meta["frames"]:
            fr["depth_file_path"] = fr["depth_file_path"].replace(".png", ".exr")
        json.dump(meta, open(jf, "w"))
    out = load_scene_processed(str(tmp_path), "dump", num_hypothesis=3)
    assert len(out) == 13 and out[10] is None and out[11] is None
    imgs, depths, valid, poses, H2, W2, intr, near, far, i_split, _, _, hyp = out
    assert (H2, W2) == (Hh, Ww) and depths.shape == (3, Hh, Ww, 1) and not valid[0, 0, 0] and valid[0, 1, 1]
    assert hyp.shape == (2, 3, Hh, Ww, 1) and hyp.min() >= near and hyp.max() <= far
    with pytest.raises(FileNotFoundError):
        load_scene_scannet(str(tmp_path), "dump", num_hypothesis=3)  # that loader takes the path literally
    sdir = os.path.join(str(tmp_path), "train", "scale_shift_inits", "s")
    os.makedirs(sdir)
    for i in range(2):
        np.save(os.path.join(sdir, f"{i}_sfminit.npy"), np.array([1.5 + i, -0.25 * i], np.float32))
    out = load_scene_processed(str(tmp_path), "dump", num_hypothesis=3, init_scales=True, scales_dir="s")
    assert len(out) == 15 and np.allclose(out[13], [1.5, 2.5]) and np.allclose(out[14], [0.0, -0.25])


def test_mean_tracker_and_image_writer(tmp_path):
    """train_utils/logging.py:5-34 and run_scade_scannet.py:396-409: 8-bit jpg, 16-bit depth png, metrics.txt."""
    from types import SimpleNamespace
    from PIL import Image
    from scade_amd.scene import MeanTracker, write_images_with_metrics
    mt = MeanTracker()
    mt.add({"psnr": 20.0, "img_loss": 0.01})
    mt.add({"psnr": 30.0, "img_loss": 0.03})
    assert mt.has("psnr") and abs(mt.get("psnr") - 25.0) < 1e-12 and abs(mt.as_dict()["img_loss"] - 0.02) < 1e-12
    g = torch.Generator().manual_seed(0)
    images = {"rgbs": torch.rand(2, 3, 6, 8, generator=g), "depths": torch.rand(2, 1, 6, 8, generator=g)}
    args = SimpleNamespace(ckpt_dir=str(tmp_path), expname="exp", scene_id="scene0758_00")
    d = write_images_with_metrics(images, mt, 5.0, args)
    assert d.endswith(os.path.join("exp", "test_images_scene0758_00"))
    assert sorted(os.listdir(d)) == ["0_d.png", "0_rgb.jpg", "1_d.png", "1_rgb.jpg", "metrics.txt"]
    dep = np.asarray(Image.open(os.path.join(d, "1_d.png")))
    assert dep.dtype == np.uint16 and dep.shape == (6, 8)
    assert np.array_equal(dep, ((2 ** 16 - 1) * images["depths"][1, 0].numpy()).astype(np.uint16))   # to16b, lossless
    rgb = np.asarray(Image.open(os.path.join(d, "0_rgb.jpg")))
    assert rgb.shape == (6, 8, 3) and rgb.dtype == np.uint8
    assert open(os.path.join(d, "metrics.txt")).read().splitlines() == ["psnr: 25.0", "img_loss: 0.02"]
    d2 = write_images_with_metrics(images, mt, 5.0, args, with_test_time_optimization=True)
    assert d2.endswith("test_images_with_optimization_scene0758_00")


def test_checkpoint_roundtrip_reference_format(tmp_path):
    import scade_amd as S
    from scade_amd.scene import load_checkpoint, restore, save_checkpoint
    torch.manual_seed(0)
    mk = lambda: S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
    c, f = mk(), mk()
    p = os.path.join(tmp_path, "exp", "100000.tar")
    save_checkpoint(p, 100000, c, f, torch.zeros(2, 1), torch.ones(2, 1))
    raw = torch.load(p, weights_only=False)
    assert all(k.startswith("module.") for k in raw["network_fn_state_dict"])          # DataParallel layout
    assert set(raw) >= {"global_step", "network_fn_state_dict", "network_fine_state_dict",
                        "optimizer_state_dict", "depth_shifts", "depth_scales"}
    c2, f2 = mk(), mk()
    step = restore(c2, f2, load_checkpoint(str(tmp_path), "exp"))
    assert step == 100000
    for a, b in zip(list(c.parameters()) + list(f.parameters()), list(c2.parameters()) + list(f2.parameters())):
        assert torch.equal(a, b)
    assert load_checkpoint(str(tmp_path), "missing") is None


def test_depth_std_map_matches_reference_formula():
    import torch
    from scade_amd.scene import depth_std_map
    g = torch.Generator().manual_seed(0)
    z = torch.rand(5, 7, 192, generator=g).sort(-1)[0] * 5
    w = torch.rand(5, 7, 192, generator=g)
    w = w / w.sum(-1, keepdim=True)
    depth = (w * z).sum(-1)
    want = ((z - depth.unsqueeze(-1)).pow(2) * w).sum(-1).clamp(0., 1.).sqrt()     # :257-258
    assert torch.equal(depth_std_map(z, w, depth), want)
    assert depth_std_map(z, w, depth).shape == (5, 7)


def test_bounding_box_cache_never_returns_a_recycled_address():
    """make_network_query_fn caches {center, scale} as one device tensor per (center, scale) OBJECT.
    Regression: the cache was once keyed on data_ptr + version, and a new tensor that landed on a freed
    tensor's address rendered with the old scene's box.  The address re-use is forced here with tensors
    that alias one numpy buffer (same data_ptr, version 0 every time)."""
    import numpy as np
    import torch
    from scade_amd.rendering import _bb_tensor
    arr_c, arr_s = np.zeros(3, dtype=np.float32), np.zeros(1, dtype=np.float32)
    for i in range(5):
        arr_c[:] = float(i)
        arr_s[:] = 0.1 * (i + 1)
        c, s = torch.from_numpy(arr_c), torch.from_numpy(arr_s)
        bb = _bb_tensor(c, s, "cpu")
        assert torch.equal(bb, torch.tensor([i, i, i, 0.1 * (i + 1)], dtype=torch.float32)), i
        assert _bb_tensor(c, s, "cpu") is bb                      # same objects: cached
        c.add_(1.0)                                               # in-place edit: new version, new box
        assert torch.equal(_bb_tensor(c, s, "cpu")[:3], c)
        del c, s, bb
