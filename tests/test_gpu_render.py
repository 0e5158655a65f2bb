"""GPU parity of render_rays end-to-end: golden fixture from the real reference
(N=32), oracle on seeded inputs, and size-independent properties at the bench size."""
import os

import numpy as np
import pytest
import torch

import scade_amd as S
from scade_amd import ops
from conftest import assert_close, load_golden, rel_l2
from oracle import scade_oracle as O
from test_oracle_golden import f6_params
from test_gpu_ops import make_net

pytestmark = pytest.mark.gpu


def build(dev, pc, pf, bbc, bbs):
    coarse, fine = make_net(pc, dev), make_net(pf, dev)
    embed_fn, _ = S.get_embedder(9, 0)
    embeddirs_fn, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(embed_fn, embeddirs_fn, bbc.to(dev), bbs.to(dev))
    return coarse, fine, query


COARSE_KEYS = ["rgb0", "disp0", "acc0", "depth0", "weights0", "z_vals0", "u"]
FINE_MAPS = ["rgb_map", "disp_map", "acc_map", "depth_map"]
FINE_REST = ["z_vals", "weights", "pred_hyp", "z_std"]


def check_ret(ret, want, tag):
    """End-to-end comparison.

    The coarse stage sees identical inputs, so it is held to the 1e-4 element-wise
    bar.  Everything after the coarse->fine resampling is compared norm-wise: the
    sample positions are an ill-conditioned function of the coarse weights (t =
    (u-c0)/(c1-c0) with tiny denominators) and the 2^8*pi positional encoding
    amplifies a 1e-7 shift of a point ~800x, so two correct fp32 implementations
    (e.g. two BLAS builds of the reference itself) differ there element-wise.  The
    fine stage is held to the element-wise bar separately, on identical inputs, by
    ``stagewise`` below.

    MEASURED (tests/test_gpu_parity64.py -> profiles/r02_parity.json, 512 rays): against an fp64
    evaluation of the same algorithm the REFERENCE's own fp32 arithmetic is off by rel-L2 2e-4
    (rgb_map), 6e-4 (depth_map), 5e-5 (z_vals), 5e-3 (weights), 1.4e-3 (pred_hyp), 6e-3 (raw); the
    HIP path is off by the same amounts (ratio 1.0-1.2) and HIP-vs-reference is of that size too.
    The bounds below therefore sit AT the fp32 noise floor of the problem for this fixture's 32
    rays, not above an achievable tighter bar."""
    assert set(want) <= set(ret)
    for k in COARSE_KEYS:
        atol = 1e-5 * float(torch.nan_to_num(want[k]).abs().max()) + 1e-7
        assert_close(ret[k], want[k], rtol=1e-4, atol=atol, what=f"{tag} {k}")
    for k in FINE_MAPS:
        e = rel_l2(torch.nan_to_num(ret[k]), torch.nan_to_num(want[k]))
        assert e < 1e-4, f"{tag} {k}: rel-L2 {e:.3e}"
    for k in FINE_REST:
        e = rel_l2(ret[k], want[k])
        assert e < 5e-4, f"{tag} {k}: rel-L2 {e:.3e}"
    psnr = -10 * torch.log10(torch.mean((ret["rgb_map"].cpu() - want["rgb_map"]) ** 2) + 1e-30)
    assert psnr > 80, f"{tag}: PSNR(build, reference) = {psnr:.1f} dB"


def stagewise(dev, want, rays, fine, query, u_fine):
    """Fine stage on the REFERENCE's intermediate tensors -> element-wise 1e-4."""
    rays = rays.to(dev)
    z = want["z_vals"].to(dev)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    raw = query(pts, rays[:, 8:11], torch.empty(0, device=dev), fine)
    assert_close(raw, want["raw"], rtol=1e-4, atol=2e-5, what="fine raw | ref z_vals")
    outs = S.raw2outputs(want["raw"].to(dev), z, rays[:, 3:6])
    for o, n in zip(outs, ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]):
        assert_close(o, want[n], rtol=1e-4, atol=1e-6, what=f"{n} | ref raw")
    zmid = .5 * (z[..., 1:] + z[..., :-1])
    s, _ = S.sample_pdf_return_u(zmid, want["weights"].to(dev)[..., 1:-1], 128, load_u=u_fine.to(dev))
    assert_close(s, want["pred_hyp"], rtol=1e-4, atol=1e-5, what="pred_hyp | ref weights")
    # coarse -> fine resampling on the reference's coarse weights, merged and sorted
    z0 = want["z_vals0"].to(dev)
    return raw


def test_render_rays_det_golden(dev):
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    with torch.no_grad():
        ret = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, embedded_cam=torch.empty(0, device=dev),
                            N_importance=128, network_fine=fine, perturb=0., retraw=True)
        want = {k[4:]: v for k, v in g.items() if k.startswith("det/")}
        assert set(want) == set(ret)
        check_ret(ret, want, "det")
        assert torch.equal(ret["z_vals0"].cpu(), want["z_vals0"])
        stagewise(dev, want, g["rays"], fine, query, want["u"])


def test_render_rays_train_forward_golden(dev):
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    with torch.no_grad():
        ret = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, embedded_cam=torch.empty(0, device=dev),
                            N_importance=128, network_fine=fine, perturb=1., retraw=True, pytest=True)
        want = {k[6:]: v for k, v in g.items() if k.startswith("train/") and k[6:] in ret}
        check_ret(ret, want, "train-fwd")
        stagewise(dev, want, g["rays"], fine, query, g["train/u"])
        il = S.img2mse(ret["rgb_map"], g["target_s"].to(dev))
        cv = S.compute_space_carving_loss(ret["pred_hyp"], g["hyp"].to(dev))
        il0 = S.img2mse(ret["rgb0"], g["target_s"].to(dev))
        # losses on the reference's own render outputs: element-wise bar
        il_r = S.img2mse(want["rgb_map"].to(dev), g["target_s"].to(dev))
        cv_r = S.compute_space_carving_loss(want["pred_hyp"].to(dev), g["hyp"].to(dev))
    assert_close(il_r, g["train/img_loss"], rtol=1e-5, atol=0, what="img_loss | ref rgb")
    assert_close(cv_r, g["train/carve"], rtol=1e-5, atol=0, what="carve | ref pred_hyp")
    assert_close(il, g["train/img_loss"], rtol=1e-4, atol=1e-7, what="img_loss")
    assert_close(cv, g["train/carve"], rtol=1e-4, atol=1e-7, what="carve")
    assert_close(il0, g["train/img_loss0"], rtol=1e-4, atol=1e-7, what="img_loss0")


def test_render_rays_vs_oracle_seeded(dev):
    N = 96
    rays = O.synthetic_rays(N, seed=21)
    pc, pf = O.nerf_init(3), O.nerf_init(4)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    with torch.no_grad():
        want = O.render_rays(rays, pc, pf, bbc, bbs, retraw=True)
    coarse, fine, query = build(dev, pc, pf, bbc, bbs)
    with torch.no_grad():
        ret = S.render_rays(rays.to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                            perturb=0., retraw=True)
        check_ret(ret, want, "seeded")
        stagewise(dev, want, rays, fine, query, want["u"])


@pytest.mark.parametrize("ns,ni", [(16, 32), (48, 200), (100, 37), (200, 320)])
def test_render_rays_other_sample_counts_vs_oracle(dev, ns, ni):
    """N_samples / N_importance other than SCADE's 64 / 128 (rows that are not multiples of the wave,
    the register sort at 64 / 256 / 512 keys, and (200, 320): a merged row of 520 keys, beyond the fused
    tail kernel, through the separate operators), jittered, with lindisp on one case.  Coarse stage:
    element-wise on identical inputs.  Every later stage: element-wise on the ORACLE's intermediates
    (the resampling is ill-conditioned - one flipped cdf bin moves a sample by a bin width - so the
    end-to-end tensors are only held to a PSNR / loose norm bar, as in check_ret)."""
    N = 24
    rays = O.synthetic_rays(N, seed=30 + ns)
    pc, pf = O.nerf_init(5), O.nerf_init(6)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    g = torch.Generator().manual_seed(ns * 1000 + ni)
    t_rand, uc, uf = torch.rand(N, ns, generator=g), torch.rand(N, ni, generator=g), torch.rand(N, ni, generator=g)
    lindisp = ns == 48
    with torch.no_grad():
        want = O.render_rays(rays, pc, pf, bbc, bbs, n_samples=ns, n_importance=ni, t_rand=t_rand,
                             u_coarse=uc, u_fine=uf, lindisp=lindisp, retraw=True)
    coarse, fine, query = build(dev, pc, pf, bbc, bbs)
    rd = rays.to(dev)
    with torch.no_grad():
        ret = S.render_rays(rd, True, coarse, query, ns, N_importance=ni, network_fine=fine,
                            perturb=1., lindisp=lindisp, t_rand=t_rand.to(dev), u_coarse=uc.to(dev),
                            cached_u=uf.to(dev))
        assert ret["z_vals"].shape == (N, ns + ni) and ret["pred_hyp"].shape == (N, ni)
        for k in ("rgb0", "depth0", "weights0", "z_vals0"):
            assert_close(ret[k], want[k], rtol=1e-4, atol=2e-6, what=k)
        # coarse tail on the oracle's coarse weights: merged z (and its points) element-wise
        z0, w0 = want["z_vals0"].to(dev), want["weights0"].to(dev)
        smp = S.sample_pdf_return_u(.5 * (z0[..., 1:] + z0[..., :-1]), w0[..., 1:-1], ni, load_u=uc.to(dev))[0]
        zm, _ = ops.merge_sorted(z0, smp, rd)
        assert_close(zm, want["z_vals"], rtol=1e-4, atol=1e-5, what="z_vals | ref coarse weights")
        # fine stage on the oracle's z_vals / raw / weights
        z = want["z_vals"].to(dev)
        pts = rd[:, None, 0:3] + rd[:, None, 3:6] * z[..., None]
        raw = query(pts, rd[:, 8:11], torch.empty(0, device=dev), fine)
        assert_close(raw, want["raw"], rtol=1e-4, atol=2e-5, what="fine raw | ref z_vals")
        if ops.ray_tail_supported(ns + ni, ni, merge=False):
            outs = ops.ray_tail(want["raw"].to(dev), z, rd, None, uf.to(dev), ni, merge=False, want_std=True)
        else:                                           # 520 samples per ray: the separate operator
            outs = S.raw2outputs(want["raw"].to(dev), z, rd[:, 3:6])
        for o, n in zip(outs[:5], ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]):
            assert_close(o, want[n], rtol=1e-4, atol=1e-6, what=f"{n} | ref raw")
        hyp = S.sample_pdf_return_u(.5 * (z[..., 1:] + z[..., :-1]), want["weights"].to(dev)[..., 1:-1], ni,
                                    load_u=uf.to(dev))[0]
        assert_close(hyp, want["pred_hyp"], rtol=1e-4, atol=1e-5, what="pred_hyp | ref weights")
    psnr = -10 * torch.log10(torch.mean((ret["rgb_map"].cpu() - want["rgb_map"]) ** 2) + 1e-30)
    assert psnr > 60, psnr
    assert rel_l2(ret["rgb_map"], want["rgb_map"]) < 1e-3 and rel_l2(ret["depth_map"], want["depth_map"]) < 1e-3


def test_render_rays_full_size_properties(dev):
    """1024 rays x (64+128): structural invariants that do not need the oracle."""
    N = 1024
    rays = O.synthetic_rays(N, seed=0).to(dev)
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    coarse, fine, query = build(dev, pc, pf, torch.zeros(3), torch.tensor(0.2))
    with torch.no_grad():
        ret = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                            perturb=1., retraw=True)
        ret2 = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                             perturb=0., retraw=True)
    for r in (ret, ret2):
        z = r["z_vals"]
        assert z.shape == (N, 192) and bool((z[:, 1:] >= z[:, :-1]).all()), "z_vals sorted"
        assert bool((r["weights"] >= 0).all()) and bool((r["acc_map"] <= 1 + 1e-5).all())
        assert_close(r["weights"].sum(-1), r["acc_map"], rtol=1e-5, atol=1e-6, what="acc == sum w")
        assert bool((r["pred_hyp"] >= z[:, :1]).all()) and bool((r["pred_hyp"] <= z[:, -1:]).all())
        assert not any(torch.isnan(v).any() for k, v in r.items() if k not in ("disp_map", "disp0"))
    # determinism of the det path
    with torch.no_grad():
        ret3 = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                             perturb=0., retraw=True)
    for k in ret2:
        assert torch.equal(torch.nan_to_num(ret2[k]), torch.nan_to_num(ret3[k])), k
    # chunking (batchify_rays) does not change results
    with torch.no_grad():
        all_ret = S.batchify_rays(rays, 300, True, network_fn=coarse, network_query_fn=query, N_samples=64,
                                  N_importance=128, network_fine=fine, perturb=0., retraw=True)
    for k in ret2:
        assert torch.equal(torch.nan_to_num(all_ret[k]), torch.nan_to_num(ret2[k])), k
    # ... nor does pipelining the chunks over two HIP streams
    with torch.no_grad():
        two = S.batchify_rays(rays, 256, True, streams=2, network_fn=coarse, network_query_fn=query,
                              N_samples=64, N_importance=128, network_fine=fine, perturb=0., retraw=True)
    torch.cuda.synchronize()
    for k in ret2:
        assert torch.equal(torch.nan_to_num(two[k]), torch.nan_to_num(ret2[k])), k


def test_render_api_unsupported_paths_fail_loudly(dev):
    rays = O.synthetic_rays(8).to(dev)
    with pytest.raises(NotImplementedError):
        S.render_rays(rays, True, None, None, 64, N_importance=0)


def test_full_image_eval_loop(dev, tmp_path):
    """render() over whole images in chunks + PSNR/RMSE on a synthetic scene
    loaded through the reference file formats."""
    from scade_amd.scene import load_scene_scannet, render_images_with_metrics, scene_bbox
    from test_scene_io_cpu import write_scene
    Hh, Ww = write_scene(str(tmp_path), Hh=20, Ww=28)
    (imgs, depths, valid, poses, _, _, intr, near, far, i_split, _, _, hyp) = \
        load_scene_scannet(str(tmp_path), "dump", num_hypothesis=3)
    t = lambda a: torch.as_tensor(a).to(dev)
    bbc, bbs = scene_bbox(Hh, Ww, intr, poses, i_split[0], far, dev)
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    coarse, fine, _ = build(dev, pc, pf, bbc.cpu(), bbs.cpu())
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(e, ed, bbc, bbs)
    kw = dict(network_fn=coarse, network_fine=fine, network_query_fn=query, N_samples=64, N_importance=128,
              perturb=False, raw_noise_std=0., use_viewdirs=True, near=near, far=far)
    res = render_images_with_metrics(t(imgs), t(depths), t(valid), t(poses), Hh, Ww, t(intr), kw, chunk=200)
    assert len(res["psnr"]) == 3 and all(np.isfinite(res["psnr"])) and res["rgbs"][0].shape == (Hh, Ww, 3)
    # the chunked full-image render equals the oracle on the same rays
    ro, rd = O.get_rays(Hh, Ww, torch.as_tensor(intr[0]), torch.as_tensor(poses[0]))
    rows = O.ray_rows(ro, rd, near, far)
    with torch.no_grad():
        want = O.render_rays(rows, pc, pf, bbc.cpu(), bbs.cpu())
    psnr = -10 * torch.log10(torch.mean((res["rgbs"][0].reshape(-1, 3).cpu() - want["rgb_map"]) ** 2) + 1e-30)
    assert psnr > 80, psnr
    want_psnr = O.mse2psnr(O.img2mse(want["rgb_map"].reshape(Hh, Ww, 3), torch.as_tensor(imgs[0])))
    assert abs(res["psnr"][0] - float(want_psnr)) < 0.05          # north_star: PSNR within 0.05 dB
    # the reference's result layout (:380-394) and its writers (:396-409, :236-264)
    from scade_amd.scene import render_video, write_images_with_metrics
    im = res["images"]
    assert im["rgbs"].shape == (3, 3, Hh, Ww) and im["depths"].shape == (3, 1, Hh, Ww) and im["rgbs0"].shape == (3, 3, Hh, Ww)
    assert im["target_valid_depths"].dtype == torch.bool and float(im["rgbs"].max()) <= 1.0
    assert torch.equal(im["depths"][0, 0], (res["depths"][0] / far).cpu())
    mm = res["mean_metrics"]
    assert abs(mm.get("psnr") - res["mean"]["psnr"]) < 1e-4 and mm.has("depth_rmse") and mm.has("psnr0")
    out_dir = write_images_with_metrics(im, mm, far, result_dir=str(tmp_path / "imgs"))
    assert len(os.listdir(out_dir)) == 7
    vdir, max_depth = render_video(t(poses), Hh, Ww, t(intr), "spiral", kw, str(tmp_path), chunk=200, run_ffmpeg=False)
    Wc = int(Hh / 9. * 16. / 3.)
    Wc -= Wc % 2
    from PIL import Image
    frames = sorted(os.listdir(vdir))
    assert frames == ["0.jpg"] and 0 < max_depth <= far + 1e-4       # every third of the three poses
    assert Image.open(os.path.join(vdir, "0.jpg")).size == (3 * Wc, Hh)   # rgb | depth | depth std


def test_render_with_5_9_is_the_centre_crop(dev):
    """render(with_5_9=True) (run_scade_scannet.py:108-115): W' = even(int(H/9*16/3)) centre columns;
    rays are independent, so the crop of the full render is the expected result bit for bit."""
    Hh, Ww = 18, 24
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    coarse, fine, query = build(dev, pc, pf, torch.zeros(3), torch.tensor(0.2))
    intr = torch.tensor([20.0, 20.0, Ww / 2, Hh / 2], device=dev)
    c2w = torch.eye(4, device=dev)[:3, :4].contiguous()
    kw = dict(chunk=128, c2w=c2w, near=0.1, far=5.0, use_viewdirs=True, network_fn=coarse, network_query_fn=query,
              N_samples=64, N_importance=128, network_fine=fine, perturb=0.)
    with torch.no_grad():
        full = S.render(Hh, Ww, intr, **kw)
        crop = S.render_hyp(Hh, Ww, intr, with_5_9=True, **kw)
    Wc = int(Hh / 9. * 16. / 3.)
    Wc -= Wc % 2
    st = (Ww - Wc) // 2
    assert crop[0].shape == (Hh, Wc, 3) and crop[3]["pred_hyp"].shape == (Hh, Wc, 128)
    assert torch.equal(crop[0], full[0][:, st:st + Wc]) and torch.equal(crop[3]["depth_map"], full[3]["depth_map"][:, st:st + Wc])
    # the explicit-rays forms of render() (:117-121) assemble the same ray rows as the c2w form
    ro, rd = S.get_rays(Hh, Ww, intr, c2w)
    kw2 = {k: v for k, v in kw.items() if k != "c2w"}
    with torch.no_grad():
        a = S.render(Hh, Ww, intr, rays=torch.stack([ro, rd]), **kw2)
        b = S.render(Hh, Ww, intr, rays=torch.stack([ro, rd, torch.zeros_like(rd)]), **kw2)
    assert a[0].shape == (Hh, Ww, 3) and torch.equal(a[0], b[0])
    assert rel_l2(a[0], full[0]) < 1e-4 and rel_l2(a[3]["depth_map"], full[3]["depth_map"]) < 1e-4


def test_fused_tails_are_bitwise_neutral(dev):
    """render_rays with the per-ray work between the MLP launches fused (scade_ray_tail, the default)
    and as separate operators: every dict entry equal bit for bit, in inference and in a recorded
    training forward, and the parameter gradients of the 3-term loss equal too."""
    N = 96
    pc, pf = O.nerf_init(2), O.nerf_init(3)
    coarse, fine, query = build(dev, pc, pf, torch.zeros(3), torch.tensor(0.2))
    rays = O.synthetic_rays(N, seed=77).to(dev)
    g = torch.Generator().manual_seed(5)
    t_rand, uc, uf = (torch.rand(N, n, generator=g).to(dev) for n in (64, 128, 128))
    tgt = torch.rand(N, 3, generator=g).to(dev)
    hyp = (torch.rand(20, N, 1, generator=g) * 4.9 + 0.1).to(dev)
    kw = dict(N_importance=128, network_fine=fine, retraw=True)
    with torch.no_grad():
        a = S.render_rays(rays, True, coarse, query, 64, perturb=0., **kw)
        b = S.render_rays(rays, True, coarse, query, 64, perturb=0., fuse_tails=False, **kw)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for every_output in (False, True):
        grads = []
        for fuse in (True, False):
            for p in list(coarse.parameters()) + list(fine.parameters()):
                p.grad = None
            r = S.render_rays(rays, True, coarse, query, 64, perturb=1., raw_noise_std=0.5, pytest=True,
                              t_rand=t_rand, u_coarse=uc, cached_u=uf, fuse_tails=fuse, **kw)
            loss = S.img2mse(r["rgb_map"], tgt) + 0.007 * S.compute_space_carving_loss(r["pred_hyp"], hyp) \
                + S.img2mse(r["rgb0"], tgt)
            if every_output:    # gradients into ALL differentiable outputs of the fused fine tail (scade_ray_tail_bwd)
                loss = loss + 1e-3 * r["weights"].pow(2).sum() + 1e-3 * r["depth_map"].sum() + 1e-3 * r["acc_map"].sum() \
                    + 1e-4 * torch.nan_to_num(r["disp_map"], posinf=0.0).clamp(max=50.).sum()
            loss.backward()
            grads.append(([p.grad.clone() for p in list(coarse.parameters()) + list(fine.parameters())], r, loss))
        for k in grads[0][1]:
            assert torch.equal(grads[0][1][k], grads[1][1][k]), k
        assert torch.equal(grads[0][2], grads[1][2])
        for x, y in zip(grads[0][0], grads[1][0]):
            assert torch.equal(x, y)


import numpy as np  # noqa: E402


def test_render_rays_ragged_batches(dev):
    """N = 1, 3, 5 rays (partial workgroups everywhere) and N = 0 (empty batch)."""
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    coarse, fine, query = build(dev, pc, pf, bbc, bbs)
    rays = O.synthetic_rays(5, seed=9)
    with torch.no_grad():
        want = O.render_rays(rays, pc, pf, bbc, bbs, retraw=True)
        for n in (1, 3, 5):
            ret = S.render_rays(rays[:n].to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                                perturb=0., retraw=True)
            for k in ("rgb0", "weights0", "depth0", "z_vals0"):
                assert_close(ret[k], want[k][:n], rtol=1e-4, atol=1e-5, what=f"N={n} {k}")
            assert rel_l2(ret["rgb_map"], want["rgb_map"][:n]) < 1e-4
        empty = S.render_rays(rays[:0].to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                              perturb=0., retraw=True)
    assert empty["rgb_map"].shape == (0, 3) and empty["z_vals"].shape == (0, 192) and empty["raw"].shape == (0, 192, 4)


def test_graphed_render_matches_eager_and_follows_weight_updates(dev):
    from scade_amd.graphs import GraphedRender
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    rays_a = g["rays"].to(dev)
    rays_b = O.synthetic_rays(32, seed=77).to(dev)
    gr = GraphedRender(32, coarse, query, 64, 128, fine)
    for rays in (rays_a, rays_b, rays_a):
        with torch.no_grad():
            want = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine, perturb=0.)
        got = gr(rays)
        for k in want:
            assert torch.equal(torch.nan_to_num(got[k]), torch.nan_to_num(want[k])), k
    with torch.no_grad():
        fine.rgb_linear.bias.add_(0.25)            # parameter update -> the graph must be re-captured
        want = S.render_rays(rays_b, True, coarse, query, 64, N_importance=128, network_fine=fine, perturb=0.)
    got = gr(rays_b)
    assert torch.equal(got["rgb_map"], want["rgb_map"])


def test_render_plumbing_golden(dev):
    """render() around render_rays (run_scade_scannet.py:80-155) against the reference's own outputs on an 18 x 40 image
    (tests/golden/f9_render_image.npz): the full image through a chunk length that does not divide it, the 5.33:9 centre
    crop, a given [2,N,3] ray batch, and c2w_staticcam (one camera's view directions on another camera's rays) - shapes,
    the split into (rgb, disp, acc, extras), and values (coarse outputs element-wise; the rest by PSNR / norm as in
    check_ret: everything behind the resampling is ill-conditioned)."""
    from test_oracle_golden import check_digest
    g = load_golden("f9_render_image")
    pc, pf = O.nerf_init(int(g["seed_coarse"])), O.nerf_init(int(g["seed_fine"]))
    check_digest(pc, g, "digest_coarse")
    check_digest(pf, g, "digest_fine")
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    Hh, Ww = int(g["H"]), int(g["W"])
    intr, c2w, c2w_b = g["intrinsic"].to(dev), g["c2w"].to(dev), g["c2w_b"].to(dev)
    kw = dict(near=0.1, far=5.0, use_viewdirs=True, network_fn=coarse, network_query_fn=query, N_samples=16,
              N_importance=24, network_fine=fine, perturb=0.0)
    with torch.no_grad():
        cases = {"full": S.render(Hh, Ww, intr, chunk=37, c2w=c2w, **kw),
                 "crop": S.render(Hh, Ww, intr, chunk=64, c2w=c2w, with_5_9=True, **kw),
                 "batch": S.render(Hh, Ww, intr, chunk=20, rays=g["batch"].to(dev), **kw),
                 "static": S.render(Hh, Ww, intr, chunk=128, c2w=c2w, c2w_staticcam=c2w_b, **kw)}
    for name, out in cases.items():
        assert len(out) == 4 and isinstance(out[3], dict), name
        rgb, disp, acc, extras = out
        assert len(extras) == int(g[f"{name}/n_extras"]), (name, sorted(extras))
        assert not ({"rgb_map", "disp_map", "acc_map"} & set(extras)), name
        for got, key in ((rgb, "rgb"), (disp, "disp"), (acc, "acc"), (extras["depth_map"], "depth_map"),
                         (extras["rgb0"], "rgb0"), (extras["z_vals"], "z_vals"), (extras["pred_hyp"], "pred_hyp")):
            want = g[f"{name}/{key}"]
            assert tuple(got.shape) == tuple(want.shape), (name, key, tuple(got.shape), tuple(want.shape))
        assert_close(extras["rgb0"], g[f"{name}/rgb0"], rtol=1e-4, atol=1e-6, what=f"{name} rgb0")
        psnr = -10 * torch.log10(torch.mean((rgb.cpu() - g[f"{name}/rgb"]) ** 2) + 1e-30)
        assert psnr > 70, (name, float(psnr))
        assert rel_l2(extras["depth_map"], g[f"{name}/depth_map"]) < 1e-3 and rel_l2(extras["z_vals"], g[f"{name}/z_vals"]) < 5e-4, name
        assert rel_l2(torch.nan_to_num(disp), torch.nan_to_num(g[f"{name}/disp"])) < 1e-3 and rel_l2(acc, g[f"{name}/acc"]) < 1e-3, name
    assert tuple(cases["crop"][0].shape) == (18, 10, 3) and tuple(cases["batch"][0].shape) == (53, 3)
