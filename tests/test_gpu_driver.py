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


def _memory_scene(Hh=40, Ww=56, n_train=4, K=6, seed=3):
    """A scene tuple held in memory (what scene.load_scene_scannet returns), smooth images + noisy hypotheses."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, Hh), np.linspace(0, 1, Ww), indexing="ij")
    n = n_train + 1
    imgs = np.stack([np.stack([xx, yy, 0.5 + 0.3 * np.sin(3 * xx + i)], -1) for i in range(n)]).astype(np.float32)
    dep = (1.0 + 1.5 * xx + 0.5 * yy).astype(np.float32)
    depths = np.repeat(dep[None, :, :, None], n, 0)
    valid = np.ones((n, Hh, Ww), bool)
    poses = np.repeat(np.eye(4, dtype=np.float32)[None], n, 0)
    poses[:, 0, 3] = np.linspace(0, 0.3, n)
    poses[:, :3, :3] += 0.01 * rng.randn(n, 3, 3).astype(np.float32)
    intr = np.repeat(np.array([[48.0, 47.0, Ww / 2, Hh / 2]], np.float32), n, 0)
    hyps = np.clip(dep[None, None, :, :, None] + 0.2 * rng.randn(n_train, K, Hh, Ww, 1).astype(np.float32), 0.1, 5.0)
    i_split = [np.arange(n_train), np.arange(0), np.arange(n_train, n), np.arange(0)]
    return (imgs, depths, valid, poses, Hh, Ww, intr, 0.1, 5.0, i_split, None, None, hyps)


@pytest.mark.parametrize("masked", [False, True])
def test_gather_batch_writes_what_get_ray_batch_returns(dev, masked):
    """scade_gather_batch (the launch in front of a captured step) against scade_gen_rays on the same pixels: same
    bits in every buffer, the view index stored, both optimizer states advanced by exactly one tick."""
    from scade_amd import ops
    from scade_amd import run_nerf_helpers as H
    data = _memory_scene()
    imgs, poses, Hh, Ww, intr, hyps = data[0], data[3], data[4], data[5], data[6], data[12]
    V, K, N = hyps.shape[0], hyps.shape[1], 96
    t_img = torch.as_tensor(imgs[:V], device=dev).contiguous()
    t_hyp = torch.as_tensor(hyps, device=dev).contiguous()
    t_pose = torch.as_tensor(poses[:V], device=dev).contiguous()
    t_intr = torch.as_tensor(intr[:V], device=dev).contiguous()
    perm = torch.randperm(Hh * Ww, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    rays, tgt, th = torch.zeros(N, 11, device=dev), torch.zeros(N, 3, device=dev), torch.zeros(K, N, 1, device=dev)
    mask = torch.full((N,), -1.0, device=dev) if masked else None
    idx = torch.full((1,), -7, device=dev, dtype=torch.long)
    st = [torch.zeros(16, device=dev), torch.zeros(16, device=dev)]
    for s_, lr in zip(st, (5e-4, 1e-7)):
        s_[1], s_[2], s_[3], s_[4], s_[5], s_[6], s_[7] = lr, 0.1, 400000.0, 0.9, 0.999, 1e-8, 1.0
    g = ops.ResidentBatchGather(Hh, Ww, t_img, t_hyp, t_pose, t_intr, 0.1, 5.0, rays, tgt, th, mask,
                                corner_px=6 if masked else 0, edge_px=2 if masked else 0, scalar_dst=idx, tick_states=st)
    off, view = 37, 2
    g(perm, off, view)
    pix = perm[off:off + N]
    coords = torch.stack([pix // Ww, pix % Ww], -1)
    # get_ray_batch's masks are the reference's fixed widths (20 / 10 px); the raw operator takes any
    want = ops.gen_rays(Hh, Ww, t_intr[view], t_pose[view], coords=coords.to(torch.int32), near=0.1, far=5.0,
                        image=t_img[view], hyps=t_hyp[view].reshape(K, Hh, Ww), corner_px=6 if masked else 0,
                        edge_px=2 if masked else 0, want_rows=True, want_mask=masked)
    assert torch.equal(rays, want["rays"]) and torch.equal(tgt, want["target_s"])
    assert torch.equal(th[..., 0], want["target_h"])
    if masked:
        assert torch.equal(mask, want["mask"]) and 0 < float(mask.sum()) < N
    assert int(idx) == view
    for s_ in st:
        assert float(s_[0]) == 1.0 and float(s_[13]) == 0.0 and abs(float(s_[9]) - 0.1) < 1e-6
    g(perm, off, view, tick_second=False)                   # the scale / shift optimizer past its freeze point
    assert float(st[0][0]) == 2.0 and float(st[1][0]) == 1.0
    with pytest.raises(IndexError):
        g(perm, off, V)
    with pytest.raises(ValueError):
        g(perm, Hh * Ww - N + 1, 0)
    with pytest.raises(ValueError):                          # a host tensor / a strided view would be a wild device read
        g(perm.cpu(), off, view)
    with pytest.raises(ValueError):
        g(torch.stack([perm, perm], -1)[:, 0], off, view)
    assert float(st[0][0]) == 2.0, "a refused call launches nothing"
    # the validated one-shot form writes the same bits
    r2, t2, h2 = torch.zeros_like(rays), torch.zeros_like(tgt), torch.zeros_like(th)
    ops.gather_batch(pix.contiguous(), Hh, Ww, t_intr[view], t_pose[view], 0.1, 5.0, t_img[view], t_hyp[view], r2, t2, h2)
    assert torch.equal(r2, rays) and torch.equal(t2, tgt) and torch.equal(h2, th)
    with pytest.raises(ValueError):
        ops.gather_batch(pix.to(torch.int32), Hh, Ww, t_intr[view], t_pose[view], 0.1, 5.0, t_img[view], t_hyp[view], r2, t2, h2)


@pytest.mark.parametrize("precision,masked", [("f32", False), ("f32", True), ("bf16-s8", False)])
def test_graphed_driver_loop_equals_the_eager_loop(dev, tmp_path, precision, masked):
    """driver.train_scene with the iteration as gather launch + graph replay against the eager get_ray_batch +
    Trainer.step loop: same views, pixels, draws and arithmetic -> the parameters, the scale / shift rows and the
    logged losses after 20 iterations agree to 1e-6 (run_scade_scannet.py:942-997)."""
    from scade_amd import driver
    data = _memory_scene()
    res = {}
    for mode in (True, False):
        torch.manual_seed(0)
        res[mode] = driver.train_scene(data, str(tmp_path / f"ck{int(mode)}"), "t", "mem", num_iterations=20, N_rand=128,
                                       i_weights=10 ** 9, i_print=5, scaleshift_lr=1e-4, test_chunk=1024, precision=precision,
                                       mask_corners=masked, log=lambda *_: None, graph=mode, no_reload=True)
    assert res[True]["graphed"] and not res[False]["graphed"]
    a, b = res[True]["trainer"], res[False]["trainer"]
    assert a.it == b.it == 20
    pa, pb = a.bucket.data, b.bucket.data
    tol = 1e-6 if precision == "f32" else 1e-5
    assert float((pa - pb).abs().max()) <= tol * max(1.0, float(pb.abs().max())), float((pa - pb).abs().max())
    assert not torch.equal(a.depth_scales, torch.ones_like(a.depth_scales)), "scale rows were stepped on"
    for (ia, la), (ib, lb) in zip(res[True]["trace"], res[False]["trace"]):
        assert ia == ib and abs(la - lb) <= 1e-5 * abs(lb) + 1e-7, (ia, la, lb)
    assert abs(res[True]["test"]["psnr"] - res[False]["test"]["psnr"]) < 1e-3


def test_graphed_driver_crosses_the_freeze_point_and_resumes(dev, tmp_path):
    """The captured loop re-captures where the reference's loop changes shape: scale / shift frozen from iteration
    freeze_ss on (:996) - their rows stop moving, the networks keep training - and a resumed run continues the
    staircase / draw streams at the checkpoint's step."""
    from scade_amd import driver
    data = _memory_scene()
    kw = dict(N_rand=64, i_weights=8, i_print=4, scaleshift_lr=1e-3, test_chunk=1024, log=lambda *_: None, freeze_ss=6)
    r = driver.train_scene(data, str(tmp_path / "ck"), "t", "mem", num_iterations=8, no_reload=True, **kw)
    tr = r["trainer"]
    assert r["graphed"] and tr.it == 8 and tr.opt.steps == 8 and tr.opt_ss.steps == 5, (tr.opt.steps, tr.opt_ss.steps)
    ss8 = tr.depth_scales.detach().clone()
    os.rename(os.path.join(str(tmp_path / "ck"), "t", "000008.tar"), os.path.join(str(tmp_path / "ck"), "t", "008000.tar"))
    r2 = driver.train_scene(data, str(tmp_path / "ck"), "t", "mem", num_iterations=12, **kw)
    assert r2["trainer"].it == 12 and torch.equal(r2["trainer"].depth_scales, ss8), "frozen rows restored and left alone"
    assert np.isfinite(r2["trace"][-1][1])


@pytest.mark.parametrize("fmt", ["f32", "bf16", "f16x3"])
def test_opening_launch_computes_the_coarse_samples_and_the_weight_packs(dev, fmt):
    """scade_gather_batch_points / scade_stage_inputs_points (round 6: the launch in front of a captured step also runs
    ray_points_draw on the batch and re-packs the networks' weight blobs in place) against the separate launches -
    scade_gather_batch / scade_stage_inputs, scade_ray_points_draw with the host's step index, scade_mlp_pack_step /
    _f16x3: every buffer bit for bit."""
    from scade_amd import ops
    from scade_amd.train import make_scade_nets
    data = _memory_scene()
    imgs, poses, Hh, Ww, intr, hyps = data[0], data[3], data[4], data[5], data[6], data[12]
    V, K, N, S, Si = hyps.shape[0], hyps.shape[1], 100, 64, 128
    t_img = torch.as_tensor(imgs[:V], device=dev).contiguous()
    t_hyp = torch.as_tensor(hyps, device=dev).contiguous()
    t_pose = torch.as_tensor(poses[:V], device=dev).contiguous()
    t_intr = torch.as_tensor(intr[:V], device=dev).contiguous()
    perm = torch.randperm(Hh * Ww, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    coarse, fine = make_scade_nets(dev, seed=21)
    prec = {"f32": "f32", "bf16": "bf16", "f16x3": "f16x3"}[fmt]
    coarse.train_precision = fine.train_precision = prec
    blob_names = {"f32": ("_packed", "_packed_t"), "bf16": ("_packed_lp", "_packed_t_lp"),
                  "f16x3": ("_packed", "_packed_f16", "_packed_t_f16")}[fmt]
    packs = ops.StepPacks([coarse, fine], fmt)
    packs.prepare()
    for n in (coarse, fine):                                  # (bytes no pack owns - slack - are 7 on both sides)
        for b in blob_names:
            getattr(n, b).fill_(7)
    ops.PARAM_EPOCH += 1
    packs.prepare()                                           # the separate launch, in place
    want_blobs = [getattr(n, b).clone() for n in (coarse, fine) for b in blob_names]
    key, step = 0x1234_5678_9ABC_DEF1, 37
    pts = ops.CoarsePoints(N, S, Si, False, dev, key=lambda: key, step=lambda: step)
    rays, tgt, th = torch.zeros(N, 11, device=dev), torch.zeros(N, 3, device=dev), torch.zeros(K, N, 1, device=dev)
    g = ops.ResidentBatchGather(Hh, Ww, t_img, t_hyp, t_pose, t_intr, 0.1, 5.0, rays, tgt, th, points=pts, packs=packs)
    for n in (coarse, fine):                                  # poison the blobs: the launch must rewrite every byte it owns
        for b in blob_names:
            getattr(n, b).fill_(7)
    off, view = 11, 1
    g(perm, off, view)
    r2, t2, h2 = torch.zeros_like(rays), torch.zeros_like(tgt), torch.zeros_like(th)
    ops.gather_batch(perm[off:off + N].contiguous(), Hh, Ww, t_intr[view], t_pose[view], 0.1, 5.0, t_img[view], t_hyp[view],
                     r2, t2, h2)
    assert torch.equal(rays, r2) and torch.equal(tgt, t2) and torch.equal(th, h2)
    z, p, ua, ub = ops.ray_points_draw(r2, S, False, ops.Draws(key, step), Si)
    assert torch.equal(pts.z, z) and torch.equal(pts.pts, p) and torch.equal(pts.u_a, ua) and torch.equal(pts.u_b, ub)
    got_blobs = [getattr(n, b) for n in (coarse, fine) for b in blob_names]
    for a, b in zip(got_blobs, want_blobs):
        assert torch.equal(a, b), "a weight blob differs from scade_mlp_pack_step's"
    # the staging form: the same coarse samples from the SOURCE rows, the same packs
    for n in (coarse, fine):
        for b in blob_names:
            getattr(n, b).fill_(7)
    pts.z.zero_(); pts.pts.zero_(); pts.u_a.zero_(); pts.u_b.zero_()
    dst = torch.zeros_like(r2)
    ops.stage_inputs([(r2, dst)], points=pts, rays=r2, packs=packs)
    assert torch.equal(dst, r2)
    assert torch.equal(pts.z, z) and torch.equal(pts.pts, p) and torch.equal(pts.u_a, ua) and torch.equal(pts.u_b, ub)
    for a, b in zip([getattr(n, b) for n in (coarse, fine) for b in blob_names], want_blobs):
        assert torch.equal(a, b)
