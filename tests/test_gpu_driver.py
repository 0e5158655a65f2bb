"""The thin driver (scade_amd/driver.py) end to end on a tiny scene written to disk in the reference's file
formats: loader -> batch gather -> Trainer.step loop -> checkpoint (reference key names) -> resume -> test
render of every test image -> images + metrics.txt on disk (run_scade_scannet.py:830-1089 call sequence)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def write_smooth_scene(root, Hh=24, Ww=32, n_train=3, n_test=1, K=4):
    """Images are smooth colour ramps (so a few dozen steps visibly lower the loss), depth maps a tilted plane,
    hypotheses = the depth plus noise."""
    from PIL import Image
    rng = np.random.RandomState(1)
    for d in ("train/rgb", "train/depth", "test/rgb", "test/depth", "train/leres_cimle/dump"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    yy, xx = np.meshgrid(np.linspace(0, 1, Hh), np.linspace(0, 1, Ww), indexing="ij")

    def frames(split, n):
        fr = []
        for i in range(n):
            rgb = np.stack([xx, yy, 0.5 + 0.3 * np.sin(3 * xx + i)], -1)
            dep = 1.0 + 1.5 * xx + 0.5 * yy
            Image.fromarray((rgb * 255).astype(np.uint8)).save(os.path.join(root, split, "rgb", f"{i}.png"))
            Image.fromarray((dep * 1000).astype(np.uint16)).save(os.path.join(root, split, "depth", f"{i}.png"))
            pose = np.eye(4)
            pose[:3, 3] = [0.05 * i, 0.0, 0.1]
            fr.append({"file_path": f"{split}/rgb/{i}.png", "depth_file_path": f"{split}/depth/{i}.png",
                       "transform_matrix": pose.tolist(), "fx": 28.0, "fy": 28.0, "cx": Ww / 2, "cy": Hh / 2})
            if split == "train":
                for j in range(K):
                    np.save(os.path.join(root, "train", "leres_cimle", "dump", f"{i}_{j}.npy"),
                            (dep + 0.2 * rng.randn(Hh, Ww)).astype(np.float32))
        return fr
    for split, n in (("train", n_train), ("test", n_test)):
        meta = {"near": 0.1, "far": 5.0, "depth_scaling_factor": 1000.0, "frames": frames(split, n)}
        json.dump(meta, open(os.path.join(root, f"transforms_{split}.json"), "w"))
    return Hh, Ww


def test_driver_trains_checkpoints_resumes_and_writes_test_images(dev, tmp_path):
    from scade_amd import driver, scene
    root, out = str(tmp_path / "scene"), str(tmp_path / "ckpt")
    Hh, Ww = write_smooth_scene(root)
    data = scene.load_scene_scannet(root, "dump", num_hypothesis=4)
    logs = []
    kw = dict(N_rand=128, i_weights=30, i_print=10, mask_corners=False, scaleshift_lr=1e-4, test_chunk=256,
              log=logs.append)
    res = driver.train_scene(data, out, "t", "tiny", num_iterations=60, i_img=30, **kw)
    assert [i for i, _ in res["val"]] == [30, 60] and res["val"][1][1]["psnr"] > res["val"][0][1]["psnr"] - 1.0, \
        "periodic validation render (:1036-1045; the test views stand in when there is no validation split)"
    trace = res["trace"]
    assert [i for i, _ in trace] == [10, 20, 30, 40, 50, 60] and all(np.isfinite(l) for _, l in trace)
    assert np.mean([l for _, l in trace[-2:]]) < 0.7 * trace[0][1], f"loss did not fall: {trace}"
    # checkpoints in the reference's format (:1006-1019: DataParallel 'module.' keys, scales / shifts)
    for it in (30, 60):
        assert os.path.exists(os.path.join(out, "t", f"{it:06d}.tar"))
    ck = torch.load(os.path.join(out, "t", "000060.tar"), weights_only=False)
    assert ck["global_step"] == 60 and "module.pts_linears.0.weight" in ck["network_fn_state_dict"]
    assert tuple(ck["depth_scales"].shape) == (3, 1) and not torch.equal(ck["depth_scales"], torch.ones(3, 1)), \
        "scale / shift rows were optimised (:954, :996)"
    # test render + writer (:1071-1086, :396-409)
    rd = os.path.join(out, "t", "test_images_tiny")
    assert sorted(os.listdir(rd)) == ["0_d.png", "0_rgb.jpg", "metrics.txt"]
    assert "psnr" in open(os.path.join(rd, "metrics.txt")).read()
    assert np.isfinite(res["test"]["psnr"]) and res["test"]["psnr"] > 10.0
    # resume: the latest '*000.tar'-style file is picked up by name; here the step count comes from '000060.tar'
    os.rename(os.path.join(out, "t", "000060.tar"), os.path.join(out, "t", "060000.tar"))
    res2 = driver.train_scene(data, out, "t", "tiny", num_iterations=70, **kw)
    assert res2["trainer"].it == 70 and [i for i, _ in res2["trace"]] == [70], "resumed at global_step 60 (:411-420)"
    assert res2["trace"][0][1] < trace[0][1], "the restored weights, not a fresh init"
    # --task test (:1262-1279): the latest checkpoint re-loaded into fresh networks gives the images of the run's end
    import shutil
    shutil.rmtree(rd)
    res_t = driver.test_scene(data, out, "t", "tiny", log=logs.append)
    assert sorted(os.listdir(rd)) == ["0_d.png", "0_rgb.jpg", "metrics.txt"]
    assert abs(res_t["mean"]["psnr"] - res["test"]["psnr"]) < 1e-3, "the step-60 checkpoint -> the test render of the 60-step run"
    # the reference's host-side pixel stream (np.random.choice without replacement) is available too
    res3 = driver.train_scene(data, str(tmp_path / "ckpt_np"), "t", "tiny", num_iterations=20, pixel_sampler="numpy", **kw)
    assert np.isfinite(res3["trace"][-1][1])
