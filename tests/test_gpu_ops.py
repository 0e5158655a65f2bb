"""GPU parity of every HIP operator against (a) the golden fixtures captured from the
real reference and (b) the CPU oracle on seeded inputs.  All calls go through the
reference-named Python API -> ctypes -> libscade_hip.so.

Tolerance (north_star): fp32 outputs within 1e-4 relative; sample_pdf index
selection bit-exact given identical (cdf, u)."""
import numpy as np
import pytest
import torch

import scade_amd as S
from scade_amd import ops
from conftest import assert_close, load_golden, rel_l2
from oracle import scade_oracle as O
from test_oracle_golden import f2_params

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=2e-6)


def make_net(params, dev):
    net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, input_ch_cam=0,
                 use_viewdirs=True)
    net.load_state_dict({k: v.detach().clone() for k, v in params.items()})
    return net.to(dev)


# ---------------------------------------------------------------- embed
def test_embed_golden(dev):
    g = load_golden("f1_embed")
    fn, dim = S.get_embedder(9, 0)
    assert dim == 57
    y = fn(g["x"].to(dev))
    assert_close(y, g["y"], rtol=1e-5, atol=2e-6, what="embed")   # |sin| <= 1: absolute 2e-6
    fn0, dim0 = S.get_embedder(0, 0)
    assert dim0 == 3 and torch.equal(fn0(g["x"].to(dev)).cpu(), g["x"])


# ---------------------------------------------------------------- MLP
def test_mlp_forward_golden(dev):
    g = load_golden("f2_mlp")
    net = make_net(f2_params(g), dev)
    with torch.no_grad():
        out = net(g["x"].to(dev))
    assert_close(out, g["out"], rtol=1e-4, atol=1e-5, what="NeRF.forward")
    assert rel_l2(out, g["out"]) < 2e-6


def test_mlp_forward_points_matches_embedded(dev):
    """mode 1 (fused embed) == mode 0 on the oracle-embedded input, ragged P (tail tile)."""
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = make_net(params, dev)
    torch.manual_seed(3)
    N, Sm = 37, 5                                     # P = 185: not a multiple of 64
    pts = torch.rand(N, Sm, 3) * 6 - 3
    vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    bbc, bbs = torch.tensor([0.1, -0.2, 0.3]), torch.tensor(0.2)
    want = O.run_network(pts, vd, lambda e: O.nerf_forward(params, e), bbc, bbs)
    embed_fn, _ = S.get_embedder(9, 0)
    embeddirs_fn, _ = S.get_embedder(0, 0)
    with torch.no_grad():
        got = S.run_network(pts.to(dev), vd.to(dev), torch.empty(0, device=dev), net, embed_fn,
                            embeddirs_fn, bbc.to(dev), bbs.to(dev))
        # generic (unfused) composition through the same operators
        got2 = S.run_network(pts.to(dev), vd.to(dev), torch.empty(0, device=dev), lambda e: net(e),
                             embed_fn, embeddirs_fn, bbc.to(dev), bbs.to(dev), netchunk=64)
    assert_close(got, want, rtol=1e-4, atol=1e-5, what="run_network fused")
    assert_close(got2, want, rtol=1e-4, atol=1e-5, what="run_network generic")


def test_mlp_edge_sizes(dev):
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = make_net(params, dev)
    for P in (0, 1, 63, 64, 65, 129):
        x = g["x"][:P] if P <= 256 else None
        with torch.no_grad():
            out = net(x.to(dev))
        assert out.shape == (P, 4)
        if P:
            assert_close(out, g["out"][:P], rtol=1e-4, atol=1e-5, what=f"P={P}")


def test_mlp_repack_after_inplace_update(dev):
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = make_net(params, dev)
    x = g["x"].to(dev)
    with torch.no_grad():
        a = net(x).clone()
        net.pts_linears[3].weight.mul_(0.5)
        b = net(x)
    params["pts_linears.3.weight"] = params["pts_linears.3.weight"] * 0.5
    assert_close(b, O.nerf_forward(params, g["x"]), rtol=1e-4, atol=1e-5, what="after in-place update")
    assert not torch.allclose(a, b)


# ---------------------------------------------------------------- composite
@pytest.mark.parametrize("Sn", [64, 192])
def test_composite_golden(dev, Sn):
    g = load_golden(f"f3_composite_{Sn}")
    raw = g["raw"].to(dev).requires_grad_(True)
    outs = S.raw2outputs(raw, g["z"].to(dev), g["d"].to(dev), 0)
    for o, n in zip(outs, ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]):
        assert_close(o, g[n], what=f"raw2outputs[{Sn}].{n}", **TOL)
    Gs = [g[k].to(dev) for k in ("G_rgb", "G_disp", "G_acc", "G_w", "G_depth")]
    # the empty ray (all sigma == 0) has disp = 1/(0/0) = NaN: keep it out of the backward seed
    Gs[1] = torch.where(torch.isnan(outs[1]), torch.zeros_like(Gs[1]), Gs[1])
    loss = sum((o * G).sum() for o, G in zip(outs, Gs) if True)
    finite = sum((torch.nan_to_num(o) * G).sum() for o, G in zip(outs, Gs))
    finite.backward()
    want = g["grad_raw"]
    ok_rows = ~torch.isnan(want).flatten(1).any(1)
    assert ok_rows.sum() >= 30
    assert_close(raw.grad.cpu()[ok_rows], want[ok_rows], rtol=2e-4, atol=1e-6, what=f"d raw2outputs[{Sn}]")
    w = S.compute_weights(raw.detach(), g["z"].to(dev), g["d"].to(dev))
    assert_close(w, g["weights"], what="compute_weights", **TOL)


def test_composite_noise_and_odd_lengths(dev):
    torch.manual_seed(5)
    for Sn in (2, 63, 65, 130, 300):   # S == 1 is degenerate in the reference (empty dists)
        N = 9
        raw = torch.randn(N, Sn, 4)
        z = torch.sort(torch.rand(N, Sn) * 4 + 0.1, -1)[0]
        d = torch.randn(N, 3)
        noise = torch.randn(N, Sn) * 0.5
        want = O.raw2outputs(raw, z, d, noise)
        got = ops.composite_fwd(raw.to(dev), z.to(dev), d.to(dev), noise.to(dev))
        for a, b, n in zip(got, want, ["rgb", "disp", "acc", "w", "depth"]):
            assert_close(a, b, what=f"S={Sn} {n}", **TOL)


# ---------------------------------------------------------------- sample_pdf
@pytest.mark.parametrize("M", [63, 191])
def test_sample_pdf_golden(dev, M):
    g = load_golden(f"f4_sample_pdf_{M}")
    bins, u = g["bins"].to(dev), g["u"].to(dev)
    N, Sn = u.shape
    # (i) bit-exact index selection given identical (cdf, u)
    s, inds, _, _ = ops.sample_pdf_fwd(bins, None, u, Sn, cdf_in=g["cdf"].to(dev), want_inds=True)
    assert torch.equal(inds.cpu(), g["inds"]), "searchsorted(right=True) indices must be bit-exact"
    assert_close(s, g["samples_u"], rtol=1e-6, atol=1e-7, what="samples from reference cdf")
    # (ii) from raw weights: cdf within 1 ulp-ish, indices equal except where u sits on a knot
    w = g["w"].to(dev).requires_grad_(True)
    s2, inds2, cdf2, _ = ops.sample_pdf_fwd(bins, w.detach(), u, Sn, want_inds=True, want_cdf=True)
    assert_close(cdf2, g["cdf"], rtol=3e-7, atol=1e-7, what="cdf")
    diff = inds2.cpu() != g["inds"]
    if diff.any():
        # every mismatch must be a u within 2 ulp of the neighbouring cdf knot
        cdf = g["cdf"]
        rr, ss = torch.nonzero(diff, as_tuple=True)
        for r, c in zip(rr.tolist(), ss.tolist()):
            k = min(int(g["inds"][r, c]), int(inds2[r, c].item()))
            assert abs(float(cdf[r, k]) - float(g["u"][r, c])) <= 2 * np.spacing(np.float32(cdf[r, k]))
    assert diff.float().mean() < 1e-3
    assert_close(s2, g["samples_u"], rtol=1e-4, atol=2e-6, what="samples from weights")
    # reference-named API: det / pytest / joint / load_u + backward
    assert_close(S.sample_pdf(bins, w, Sn, det=True), g["samples_det"], rtol=1e-4, atol=2e-6, what="det")
    assert_close(S.sample_pdf(bins, w, Sn, det=False, pytest=True), g["samples_pytest"], rtol=1e-4,
                 atol=2e-6, what="pytest")
    sj, uj = S.sample_pdf_joint_return_u(bins, w, Sn, load_u=g["u_joint"].to(dev).expand(N, Sn))
    assert_close(sj, g["samples_joint"], rtol=1e-4, atol=2e-6, what="joint")
    su, ub = S.sample_pdf_return_u(bins, w, Sn, det=False, load_u=u)
    assert ub is u
    (su * g["G"].to(dev)).sum().backward()
    assert_close(w.grad, g["grad_w"], rtol=2e-3, atol=2e-4 * float(g["grad_w"].abs().max()), what="grad w")
    assert rel_l2(w.grad, g["grad_w"]) < 1e-4


def test_sample_pdf_random_draws_are_valid(dev):
    torch.manual_seed(0)
    bins = torch.sort(torch.rand(16, 63) * 4 + 0.1, -1)[0].to(dev)
    w = torch.rand(16, 62).to(dev)
    s = S.sample_pdf(bins, w, 128, det=False)
    assert s.shape == (16, 128)
    assert bool((s >= bins[:, :1]).all()) and bool((s <= bins[:, -1:]).all())
    s2, u = S.sample_pdf_joint_return_u(bins, w, 128, det=False)
    assert u.shape == (16, 128) and bool((u[0] == u[5]).all())


# ---------------------------------------------------------------- merge
def test_merge_sorted(dev):
    torch.manual_seed(1)
    N = 33
    za = torch.sort(torch.rand(N, 64) * 5, -1)[0]
    zb = torch.rand(N, 128) * 5
    zb[0, :10] = za[0, :10]                                  # exact ties
    rays = O.synthetic_rays(N, seed=4, unit_dirs=False)
    want, _ = torch.sort(torch.cat([za, zb], -1), -1)
    got, pts = ops.merge_sorted(za.to(dev), zb.to(dev), rays.to(dev))
    assert torch.equal(got.cpu(), want)
    wpts = rays[:, None, 0:3] + rays[:, None, 3:6] * want[..., None]
    assert torch.equal(pts.cpu(), wpts)                      # mul then add, no FMA contraction


# ---------------------------------------------------------------- ray_points / perturb
def test_ray_points_and_perturb(dev):
    g = load_golden("f7_perturb")
    rays = O.synthetic_rays(32, seed=2, unit_dirs=False)
    z, pts = ops.ray_points(rays.to(dev), 64, None, False)
    assert torch.equal(z.cpu(), g["z"])
    zp, pts2 = ops.ray_points(rays.to(dev), 64, g["t_rand"].to(dev), False)
    assert torch.equal(zp.cpu(), g["out"])
    assert torch.equal(pts2.cpu(), rays[:, None, 0:3] + rays[:, None, 3:6] * g["out"][..., None])
    np.random.seed(0)
    assert torch.equal(S.perturb_z_vals(g["z"].to(dev), True).cpu(), g["out"])
    # lindisp
    zl, _ = ops.ray_points(rays.to(dev), 64, None, True)
    t = torch.linspace(0., 1., 64)
    want = 1. / (1. / rays[:, 6:7] * (1. - t) + 1. / rays[:, 7:8] * t)
    assert_close(zl, want, rtol=1e-6, atol=0, what="lindisp")


# ---------------------------------------------------------------- carve / mse
@pytest.mark.parametrize("K", [20, 40])
def test_carve_golden(dev, K):
    g = load_golden(f"f5_carve_{K}")
    mask = g["mask"].to(dev)
    variants = {"default": {}, "mask": dict(mask=mask), "thr": dict(threshold=0.05),
                "joint": dict(is_joint=True), "p1": dict(norm_p=1),
                "mask_thr": dict(mask=mask, threshold=0.05)}
    for name, kw in variants.items():
        p = g["pred"].to(dev).requires_grad_(True)
        h = g["hyp"].to(dev).requires_grad_(True)
        loss = S.compute_space_carving_loss(p, h, **kw)
        (loss * 1.5).backward()
        assert_close(loss, g[f"{name}/loss"], rtol=1e-5, atol=1e-7, what=f"carve {name}")
        assert_close(p.grad, 1.5 * g[f"{name}/grad_pred"], rtol=1e-5, atol=1e-9, what=f"{name} dpred")
        assert_close(h.grad, 1.5 * g[f"{name}/grad_hyp"], rtol=1e-4, atol=1e-8, what=f"{name} dhyp")


def test_space_carving_cached_quantile_hypotheses_golden(dev):
    """target_hypothesis [K,N,P]: every sample already picked its hypothesis (helpers:100-102) - loss and
    both gradients against the reference, plain and ray-sharded entry."""
    g = load_golden("f5_carve_knp")
    mask = g["mask"].to(dev)
    variants = {"default": {}, "mask": dict(mask=mask), "thr": dict(threshold=0.05),
                "joint": dict(is_joint=True), "joint_mask_thr": dict(is_joint=True, mask=mask, threshold=0.05),
                "mask_thr": dict(mask=mask, threshold=0.05)}
    for name, kw in variants.items():
        for sharded in ((False, True) if kw.get("is_joint") else (False,)):
            p = g["pred"].to(dev).requires_grad_(True)
            h = g["hyp"].to(dev).requires_grad_(True)
            loss = S.compute_space_carving_loss(p, h, sharded=sharded, **kw)
            (loss * 1.5).backward()
            assert_close(loss, g[f"{name}/loss"], rtol=1e-5, atol=1e-7, what=f"carve knp {name}")
            assert_close(p.grad, 1.5 * g[f"{name}/grad_pred"], rtol=1e-5, atol=1e-9, what=f"knp {name} dpred")
            assert_close(h.grad, 1.5 * g[f"{name}/grad_hyp"], rtol=1e-5, atol=1e-9, what=f"knp {name} dhyp")
    with pytest.raises(ValueError):
        S.compute_space_carving_loss(g["pred"].to(dev), g["hyp"].to(dev)[:, :, :7])


def test_mse(dev):
    torch.manual_seed(2)
    x, y, m = torch.rand(100, 3), torch.rand(100, 3), (torch.rand(100) > 0.5).float()
    xd = x.to(dev).requires_grad_(True)
    l = S.img2mse(xd, y.to(dev))
    l.backward()
    xr = x.clone().requires_grad_(True)
    lr = O.img2mse(xr, y)
    lr.backward()
    assert_close(l, lr, rtol=1e-6, atol=0, what="mse")
    assert_close(xd.grad, xr.grad, rtol=1e-6, atol=1e-10, what="dmse")
    lm = S.img2mse_masked(x.to(dev), y.to(dev), m.to(dev))
    assert_close(lm, torch.mean((x - y) ** 2 * m[:, None]), rtol=1e-6, atol=0, what="masked mse")
    assert_close(S.mse2psnr(l.detach()), O.mse2psnr(lr.detach()), rtol=1e-6, atol=0, what="psnr")


# ---------------------------------------------------------------- ray generation / batch gather
def test_gen_rays_golden(dev):
    g = load_golden("f8_rays")
    Hh, Ww = int(g["H"]), int(g["W"])
    intr, c2w = g["intrinsic"].to(dev), g["c2w"].to(dev)
    ro, rd = S.get_rays(Hh, Ww, intr, c2w)
    assert ro.shape == (Hh, Ww, 3)
    assert_close(rd, g["rays_d_full"], rtol=2e-6, atol=1e-7, what="get_rays d")
    assert torch.equal(ro.cpu(), g["rays_o_full"])
    sel = g["sel"]
    ro2, rd2 = S.get_rays(Hh, Ww, intr, c2w, coords=sel.float().to(dev))
    assert_close(rd2, g["rays_d_full"][sel[:, 0], sel[:, 1]], rtol=2e-6, atol=1e-7, what="get_rays coords")
    rays, ts, th, mask = S.get_ray_batch(Hh, Ww, intr, c2w, sel.to(dev), float(g["near"]), float(g["far"]),
                                        image=g["image"].to(dev), hypotheses=g["hyps"].to(dev),
                                        mask_corners=True)
    assert_close(rays, g["rows"], rtol=2e-6, atol=1e-7, what="ray rows")
    assert torch.equal(ts.cpu(), g["target_s"]) and torch.equal(th.cpu(), g["target_h"])
    want_mask = torch.ones(Hh, Ww)
    want_mask[:20, :20] = 0; want_mask[:20, -20:] = 0; want_mask[-20:, :20] = 0; want_mask[-20:, -20:] = 0
    assert torch.equal(mask.cpu(), want_mask[sel[:, 0], sel[:, 1]])
    # full-image render() path builds the same rows
    rows_full = ops.gen_rays(Hh, Ww, intr, c2w, near=0.1, far=5.0)["rays"]
    want = O.ray_rows(g["rays_o_full"], g["rays_d_full"], 0.1, 5.0)
    assert_close(rows_full, want, rtol=2e-6, atol=1e-7, what="full rows")


def test_space_carving_joint_sharded_entry_single_process(dev):
    """sharded=True with one process: the two-phase entry (column means | exchange | min) gives
    the same loss and gradients as the fused joint kernel and the oracle."""
    g = torch.Generator().manual_seed(21)
    N, K, P = 37, 7, 128
    pred = torch.rand(N, P, generator=g) * 4 + 0.5
    hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
    mask = (torch.rand(N, generator=g) > 0.3).float()
    for m, thr in ((None, 0.0), (mask, 0.05)):
        po, ho = pred.clone().requires_grad_(True), hyp.clone().requires_grad_(True)
        want = O.compute_space_carving_loss(po, ho, is_joint=True, mask=m, threshold=thr)
        want.backward()
        res = []
        for sharded in (False, True):
            p = pred.to(dev).requires_grad_(True)
            h = hyp.to(dev).requires_grad_(True)
            l = S.compute_space_carving_loss(p, h, is_joint=True, mask=None if m is None else m.to(dev),
                                             threshold=thr, sharded=sharded)
            l.backward()
            res.append((l.detach().cpu(), p.grad.cpu(), h.grad.cpu()))
        for l, gp, gh in res:
            assert_close(l, want.detach(), rtol=1e-5, atol=1e-7, what="joint loss")
            assert_close(gp, po.grad, rtol=1e-5, atol=1e-9, what="joint d/d pred")
            assert_close(gh, ho.grad, rtol=1e-5, atol=1e-9, what="joint d/d hyp")


# ---------------------------------------------------------------- randomized shape sweep
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_per_ray_kernels_random_shapes_forward_and_backward(dev, seed):
    """Shape sweep (ragged ray counts, sample counts on both sides of the 64-lane wave, sampler
    bins that are not 63 / 191, hypothesis counts that are not 20 / 40) of the per-ray kernels,
    forward AND backward, against the oracle and its autograd."""
    g = torch.Generator().manual_seed(100 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for _ in range(4):
        N, Sn, Ns, K = ri(1, 70), ri(2, 260), ri(1, 150), ri(1, 45)
        # --- compositing
        raw = torch.randn(N, Sn, 4, generator=g)
        z = torch.sort(torch.rand(N, Sn, generator=g) * 4 + 0.1, -1)[0]
        d = torch.randn(N, 3, generator=g)
        ro = raw.clone().requires_grad_(True)
        wo = O.raw2outputs(ro, z, d)
        cots = [torch.randn(t.shape, generator=g) for t in wo]
        sum((a * c).sum() for a, c in zip(wo, cots)).backward()
        rg = raw.to(dev).requires_grad_(True)
        got = S.raw2outputs(rg, z.to(dev), d.to(dev))
        sum((a * c.to(dev)).sum() for a, c in zip(got, cots)).backward()
        for a, b, n in zip(got, wo, ["rgb", "disp", "acc", "w", "depth"]):
            assert_close(a, b.detach(), what=f"composite N={N} S={Sn} {n}", **TOL)
        assert_close(rg.grad, ro.grad, rtol=1e-4, atol=1e-5 * float(ro.grad.abs().max()), what=f"composite grad N={N} S={Sn}")
        # --- sampler (forward from weights with explicit u, backward w.r.t. the weights)
        M = max(Sn - 1, 2)
        bins = torch.sort(torch.rand(N, M, generator=g) * 4 + 0.1, -1)[0]
        w = torch.rand(N, M - 1, generator=g) ** 4
        if N > 2:
            w[0] = 0.0                                           # all-zero row
            w[1] = 0.0
            w[1, (M - 1) // 2] = 1.0                             # one-hot row
        u = torch.rand(N, Ns, generator=g)
        wo_ = w.clone().requires_grad_(True)
        so = O.sample_pdf(bins, wo_, u)
        cot = torch.randn(so.shape, generator=g)
        (so * cot).sum().backward()
        wg = w.to(dev).requires_grad_(True)
        sg, _ = S.sample_pdf_return_u(bins.to(dev), wg, Ns, load_u=u.to(dev))
        (sg * cot.to(dev)).sum().backward()
        # from raw weights the two cdfs differ by an ulp (torch.sum's vectorised cascade is not
        # reproducible), so a u within an ulp of a cdf knot may select the neighbouring bin: allow a
        # vanishing fraction of such samples (the golden test pins the indices given identical cdf, u)
        err = (sg.cpu() - so.detach()).abs()
        bad = err > 2e-5 + 1e-4 * so.detach().abs()
        assert float(bad.float().mean()) < 1e-3, (N, M, Ns, int(bad.sum()))
        assert float(err.max()) < 0.5
        scale = float(wo_.grad.abs().max()) + 1e-30
        assert rel_l2(wg.grad, wo_.grad) < 2e-3, (N, M, Ns, rel_l2(wg.grad, wo_.grad))   # 1/den^2 amplification
        # --- merge
        zb = torch.rand(N, Ns, generator=g) * 5
        rays = O.synthetic_rays(N, seed=200 + seed, unit_dirs=False)
        zs, pts = ops.merge_sorted(z.to(dev), zb.to(dev), rays.to(dev))
        assert torch.equal(zs.cpu(), torch.sort(torch.cat([z, zb], -1), -1)[0])
        # --- space carving
        pred = torch.rand(N, Ns, generator=g) * 5
        hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
        po, ho = pred.clone().requires_grad_(True), hyp.clone().requires_grad_(True)
        lo = O.compute_space_carving_loss(po, ho)
        lo.backward()
        pg, hg = pred.to(dev).requires_grad_(True), hyp.to(dev).requires_grad_(True)
        lg = S.compute_space_carving_loss(pg, hg)
        lg.backward()
        assert_close(lg, lo.detach(), rtol=1e-5, atol=1e-7, what=f"carve N={N} P={Ns} K={K}")
        assert_close(pg.grad, po.grad, rtol=1e-5, atol=1e-9, what="carve d/d pred")
        assert_close(hg.grad, ho.grad, rtol=1e-5, atol=1e-9, what="carve d/d hyp")


def test_merge_sorted_sizes_nan_and_unsorted_inputs(dev):
    """Register network (rows up to 512 keys), LDS network (larger), ties, unsorted first operand,
    NaN placed last like torch.sort."""
    g = torch.Generator().manual_seed(12)
    for N, Sa, Sb in ((5, 1, 1), (9, 7, 5), (33, 64, 128), (17, 64, 0), (4, 100, 156), (3, 200, 312),
                      (2, 300, 500), (2, 1000, 1500)):
        za = torch.rand(N, Sa, generator=g) * 5                      # deliberately NOT sorted
        zb = torch.rand(N, Sb, generator=g) * 5
        if Sb > 3:
            zb[0, :3] = za[0, :1]                                    # ties
            zb[-1, 1] = float("nan")
        rays = O.synthetic_rays(N, seed=4, unit_dirs=False)
        want, _ = torch.sort(torch.cat([za, zb], -1), -1)
        got, pts = ops.merge_sorted(za.to(dev), zb.to(dev), rays.to(dev))
        assert torch.equal(torch.nan_to_num(got.cpu(), nan=-7.0), torch.nan_to_num(want, nan=-7.0)), (N, Sa, Sb)
        assert torch.equal(torch.isnan(got.cpu()), torch.isnan(want))
        wpts = rays[:, None, 0:3] + rays[:, None, 3:6] * want[..., None]
        assert torch.equal(torch.nan_to_num(pts.cpu()), torch.nan_to_num(wpts))


def test_ray_tail_equals_the_three_operators(dev):
    """scade_ray_tail (raw2outputs -> sample_pdf on z_mid / weights[1:-1] -> sorted merge -> points in
    one launch) against the separate entries on the same inputs: every output bit for bit, over ragged
    sizes, density noise, a shared u row, and a NaN density."""
    g = torch.Generator().manual_seed(21)
    for N, S, Si, merge in ((7, 64, 128, True), (5, 3, 1, True), (9, 40, 17, True), (6, 130, 100, True),
                            (4, 256, 256, True), (3, 192, 128, False), (5, 300, 64, False), (2, 512, 1000, False)):
        raw = torch.randn(N, S, 4, generator=g)
        z = torch.sort(torch.rand(N, S, generator=g) * 4.9 + 0.1, -1).values
        rays = O.synthetic_rays(N, seed=N + S, unit_dirs=False)
        noise = torch.randn(N, S, generator=g) * 0.3 if S % 2 == 0 else None
        u = torch.rand(Si, generator=g) if N == 9 else torch.rand(N, Si, generator=g)
        if N == 6:
            raw[1, 5, 3] = float("nan")
        a = [t.to(dev) if t is not None else None for t in (raw, z, rays, noise, u)]
        assert ops.ray_tail_supported(S, Si, merge)
        got = ops.ray_tail(a[0], a[1], a[2], a[3], a[4], Si, merge=merge, want_std=True)
        rgb, disp, acc, w, depth = ops.composite_fwd(a[0], a[1], a[2][:, 3:6], a[3])
        smp, _, _, std = ops.sample_pdf_fwd(a[1], w[:, 1:-1], a[4], Si, bins_are_mids=True, want_std=True)
        want = [rgb, disp, acc, w, depth, smp, std]
        if merge:
            want += list(ops.merge_sorted(a[1], smp, a[2]))
        eq = lambda x, y: torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0))
        for k, (x, y) in enumerate(zip(got, want)):
            assert eq(x, y), (N, S, Si, merge, k)
    assert not ops.ray_tail_supported(300, 300, True) and not ops.ray_tail_supported(2, 8, False)
    with pytest.raises(RuntimeError):
        ops.ray_tail(torch.zeros(2, 300, 4, device=dev), torch.zeros(2, 300, device=dev),
                     torch.zeros(2, 11, device=dev), None, torch.zeros(2, 300, device=dev), 300, merge=True)


def test_mlp_point_tilings_agree(dev):
    """The exact forward / dgrad kernels run 64-point workgroups, or 32-point ones when the launch is too
    small to fill the chip (pick_point_tiles: P < 32768 on a 256-CU part).  The golden fixtures pin the
    small launches; this ties the large ones to them: per-point outputs are bit-identical across the two
    tilings (same k order), gradients agree to summation order."""
    net = make_net(O.nerf_init(3), dev)
    g = torch.Generator().manual_seed(9)
    P, Ps = 40000, 3000
    x = torch.cat([O.embed(torch.rand(P, 3, generator=g) * 2 - 1, 9),
                   torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)], -1).to(dev)
    with torch.no_grad():
        big, small = net(x), net(x[:Ps])
    assert torch.equal(big[:Ps], small)
    G = torch.zeros(P, 4)
    G[:Ps] = torch.randn(Ps, 4, generator=g) * 1e-3
    G = G.to(dev)
    grads = []
    for xs, gs in ((x, G), (x[:Ps], G[:Ps])):
        for p in net.parameters():
            p.grad = None
        net(xs).backward(gs)
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
    assert rel_l2(grads[0], grads[1]) < 1e-5


def test_wrong_current_device_fails_loudly(dev):
    """Kernels go to the current device's stream; an operator entry given a tensor of ANOTHER device
    raises instead of dereferencing it on the wrong GPU (one GPU here: checked through the guard)."""
    from scade_amd._lib import check_current_device
    t = torch.zeros(4, device=dev)
    check_current_device(t, "x")                          # same device: fine
    real = torch.cuda.current_device
    torch.cuda.current_device = lambda: 3                 # pretend another device is current
    try:
        with pytest.raises(RuntimeError, match="current device"):
            check_current_device(t, "x")
        with pytest.raises(RuntimeError, match="current device"):
            S.render_rays(torch.zeros(2, 11, device=dev), True, None, None, 64, N_importance=128)
    finally:
        torch.cuda.current_device = real


def _philox4x32_10(counter, key):
    """Reference Philox4x32-10 (Salmon et al., SC'11) on numpy uint32 arrays: counter [...,4], key (k0, k1)."""
    import numpy as np
    c = counter.astype(np.uint64)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[..., 0], M1 * c[..., 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[..., 1] ^ k0) & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[..., 3] ^ k1) & MASK
        c = np.stack([n0, p1 & MASK, n2, p0 & MASK], -1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c.astype(np.uint32)


def test_philox_reference_known_answer():
    """The numpy reference above against Random123's published known-answer vectors for philox4x32-10."""
    import numpy as np
    z = _philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = _philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


@pytest.mark.parametrize("S,Si", [(64, 128), (7, 5), (130, 33)])
def test_ray_points_draw_is_philox_keyed_by_seed_step_ray(dev, S, Si):
    """scade_ray_points_draw: the step's uniform draws made inside the kernel are Philox4x32-10 with key = seed,
    counter = (draw block, ray, step lo, step hi); per ray [S jitter | Si | Si] with each part padded to blocks
    of four; value = (bits >> 8) * 2^-24.  Checked bit for bit against the numpy reference: the two sampler rows
    directly, the jitter through scade_ray_points fed with the reference draws (same z, same points).  The device
    resident step (graph-captured steps) gives the same stream as the host step."""
    import numpy as np
    from scade_amd import ops
    N = 37
    rays = O.synthetic_rays(N, seed=3).to(dev)
    seed, step = 0x0123456789ABCDEF, 2 ** 33 + 5
    jb, sb = (S + 3) // 4, (Si + 3) // 4
    nblk = jb + 2 * sb
    ctr = np.zeros((N, nblk, 4), np.uint32)
    ctr[..., 0] = np.arange(nblk)[None]
    ctr[..., 1] = np.arange(N)[:, None]
    ctr[..., 2] = step & 0xFFFFFFFF
    ctr[..., 3] = step >> 32
    bits = _philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    u = ((bits >> 8).astype(np.float32) * np.float32(2.0 ** -24)).reshape(N, nblk * 4)
    want_t, want_a, want_b = u[:, :S], u[:, 4 * jb:4 * jb + Si], u[:, 4 * (jb + sb):4 * (jb + sb) + Si]
    z, pts, ua, ub = ops.ray_points_draw(rays, S, False, ops.Draws(seed, step), Si)
    assert torch.equal(ua.cpu(), torch.from_numpy(want_a.copy())) and torch.equal(ub.cpu(), torch.from_numpy(want_b.copy()))
    z_ref, pts_ref = ops.ray_points(rays, S, torch.from_numpy(want_t.copy()).to(dev), False)
    assert torch.equal(z, z_ref) and torch.equal(pts, pts_ref)
    assert 0.0 <= float(ua.min()) and float(ua.max()) < 1.0
    # another step / another seed: other draws; only one array wanted: the other is not touched
    z2, _, ua2, none_b = ops.ray_points_draw(rays, S, False, ops.Draws(seed, step + 1), Si, want_b=False)
    assert none_b is None and not torch.equal(ua2, ua) and not torch.equal(z2, z)
    # device-resident step count (a float, as FusedAdam.state[0] holds it)
    sd = torch.tensor([12345.0], device=dev)
    a = ops.ray_points_draw(rays, S, False, ops.Draws(seed, 999, step_dev=sd), Si)
    b = ops.ray_points_draw(rays, S, False, ops.Draws(seed, 12345), Si)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_in_kernel_draws_are_uniform(dev):
    from scade_amd import ops
    rays = O.synthetic_rays(2048, seed=4).to(dev)
    _, _, ua, ub = ops.ray_points_draw(rays, 64, False, ops.Draws(7, 3), 128)
    x = torch.cat([ua.flatten(), ub.flatten()]).double()
    assert abs(float(x.mean()) - 0.5) < 2e-3 and abs(float(x.var()) - 1 / 12) < 1e-3
    hist = torch.histc(x.float(), bins=16, min=0, max=1) / x.numel()
    assert float((hist - 1 / 16).abs().max()) < 2e-3
    assert abs(float(torch.corrcoef(torch.stack([ua.flatten(), ub.flatten()]))[0, 1])) < 5e-3


def test_stage_inputs_copies_every_pair_and_the_scalar_in_one_launch(dev):
    """scade_stage_inputs: up to eight device copies + one int64 scalar; 16-byte fast path and the 4-byte path for
    a destination that is not 16-byte aligned; unsupported pairs fall back to copy_."""
    from scade_amd import ops
    g = torch.Generator().manual_seed(3)
    srcs = [torch.rand(n, generator=g).to(dev) for n in (1, 5, 1024 * 11 + 3, 4096, 7)]
    big = torch.zeros(1024 * 11 + 3 + 1, device=dev)
    dsts = [torch.zeros_like(srcs[0]), torch.zeros_like(srcs[1]), big[1:], torch.zeros_like(srcs[3]), torch.zeros(7, device=dev, dtype=torch.float64)]
    idx = torch.zeros(1, device=dev, dtype=torch.long)
    ops.stage_inputs(list(zip(srcs, dsts)), scalar=(idx, 41))
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(d.float(), s)
    assert float(big[0]) == 0.0 and int(idx) == 41
    # a pair that already is the destination costs nothing; nothing to do at all is fine too
    ops.stage_inputs([(dsts[0], dsts[0])])
    ops.stage_inputs([], scalar=(idx, 7))
    assert int(idx) == 7


def test_sample_pdf_without_a_weight_is_refused_like_the_reference(dev):
    """One bin edge / zero weights (N_samples = 2 coarse samples): the reference's cdf comes out empty (helpers:342-343)
    and its gather raises (:373); the operator says why instead of handing the C ABI an empty tensor."""
    bins, w, u = torch.rand(4, 1, device=dev), torch.zeros(4, 0, device=dev), torch.rand(4, 8, device=dev)
    with pytest.raises(ValueError, match="at least two"):
        S.sample_pdf_return_u(bins, w, 8, load_u=u)
    with pytest.raises((RuntimeError, IndexError)):
        O.sample_pdf(bins.cpu(), w.cpu(), u.cpu())


@pytest.mark.parametrize("special", [float("nan"), float("inf"), -float("inf"), 0.0, 1e-45, 1e38, -1e38])
def test_special_values_leave_the_same_nan_and_inf_patterns_as_the_oracle(dev, special):
    """A NaN / Inf / denormal / huge entry of the network output, of a sampler weight, of a depth hypothesis or of a
    colour lands where torch leaves it: the positions of NaN, +Inf and -Inf of every forward output agree with the oracle's
    (F.relu and torch.min PROPAGATE NaN - v_min_f32 / fminf would return the other operand), and so does the compositing
    gradient (threshold_backward passes a NaN through).  Gradients of the space-carving loss w.r.t. NON-FINITE inputs are
    not compared: torch's norm backward turns inf / inf into NaN there, the kernels return the sign."""
    g = torch.Generator().manual_seed(5)
    N, Sn = 12, 70

    def same_pattern(a, b, what):
        a, b = a.detach().cpu(), b.detach()
        assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: NaN positions"
        assert torch.equal(torch.isposinf(a), torch.isposinf(b)) and torch.equal(torch.isneginf(a), torch.isneginf(b)), f"{what}: Inf positions"
        fin = torch.isfinite(b)
        if bool(fin.any()):
            assert_close(a[fin], b[fin], rtol=2e-4, atol=1e-5 * float(b[fin].abs().max()) + 1e-9, what=what)

    raw = torch.randn(N, Sn, 4, generator=g)
    z = torch.sort(torch.rand(N, Sn, generator=g) * 4 + 0.1, -1)[0]
    d = torch.randn(N, 3, generator=g)
    for ch in range(4):
        raw[ch, 5 + ch, ch] = special
    raw[8, Sn - 1, 3] = special
    ro = raw.clone().requires_grad_(True)
    wo = O.raw2outputs(ro, z, d)
    sum(t.sum() for t in wo).backward()
    rg = raw.to(dev).requires_grad_(True)
    got = S.raw2outputs(rg, z.to(dev), d.to(dev))
    sum(t.sum() for t in got).backward()
    for a, b, n in zip(got, wo, ["rgb", "disp", "acc", "w", "depth"]):
        same_pattern(a, b, f"composite {n}")
    same_pattern(rg.grad, ro.grad, "composite gradient")
    # sampler
    M = Sn - 1
    bins = torch.sort(torch.rand(N, M, generator=g) * 4 + 0.1, -1)[0]
    w = torch.rand(N, M - 1, generator=g)
    w[2, 7] = special
    w[5, 0] = special
    u = torch.rand(N, 33, generator=g)
    so = O.sample_pdf(bins, w, u)
    sg, _ = S.sample_pdf_return_u(bins.to(dev), w.to(dev), 33, load_u=u.to(dev))
    same_pattern(sg, so, "sampler")
    # space carving and the photometric term
    pred = torch.rand(N, 20, generator=g) * 5
    hyp = torch.rand(7, N, 1, generator=g) * 4.9 + 0.1
    pred[1, 3] = special
    hyp[2, 4, 0] = special
    same_pattern(S.compute_space_carving_loss(pred.to(dev), hyp.to(dev)).reshape(1),
                 O.compute_space_carving_loss(pred, hyp).reshape(1), "carve loss")
    x, y = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
    x[0, 1] = special
    xo = x.clone().requires_grad_(True)
    O.img2mse(xo, y).backward()
    xg = x.to(dev).requires_grad_(True)
    mg = S.img2mse(xg, y.to(dev))
    mg.backward()
    same_pattern(mg.reshape(1), O.img2mse(x, y).reshape(1), "mse")
    same_pattern(xg.grad, xo.grad, "mse gradient")


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_space_carving_variants_random_shapes_forward_and_backward(dev, seed):
    """Every variant of the space-carving loss (per-ray / joint minimum, mask, threshold, hypotheses per ray [K,N,1] or
    per sample [K,N,P]) at random ragged shapes - hypothesis counts on both sides of the 64-lane wave, one ray, one
    sample, one hypothesis - forward and backward against the oracle and its autograd."""
    g = torch.Generator().manual_seed(300 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for it in range(6):
        N, P, K = ri(1, 90), ri(1, 210), ri(1, 80)
        if it == 0:
            N, P, K = 1, 1, 1
        pred = torch.rand(N, P, generator=g) * 5
        for per_sample in (False, True):
            hyp = torch.rand(K, N, P if per_sample else 1, generator=g) * 4.9 + 0.1
            mask = (torch.rand(N, generator=g) > 0.3).float()
            for kw in (dict(), dict(is_joint=True), dict(mask=mask), dict(threshold=0.3), dict(is_joint=True, mask=mask, threshold=0.3)):
                po, ho = pred.clone().requires_grad_(True), hyp.clone().requires_grad_(True)
                lo = O.compute_space_carving_loss(po, ho, **kw)
                lo.backward()
                pg, hg = pred.to(dev).requires_grad_(True), hyp.to(dev).requires_grad_(True)
                kd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
                lg = S.compute_space_carving_loss(pg, hg, **kd)
                lg.backward()
                tag = f"N={N} P={P} K={K} per_sample={per_sample} {sorted(kw)}"
                assert_close(lg, lo.detach(), rtol=1e-5, atol=1e-7, what="carve " + tag)
                assert_close(pg.grad, po.grad, rtol=1e-5, atol=1e-9, what="carve d/d pred " + tag)
                assert_close(hg.grad, ho.grad, rtol=1e-5, atol=1e-9, what="carve d/d hyp " + tag)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_ray_batch_random_cameras_and_masks_vs_oracle(dev, seed):
    """get_rays / get_ray_batch for random image sizes (down to images SMALLER than the 20 px corner and 10 px edge bands),
    intrinsics, poses and pixel selections that include the four corner pixels - against the oracle's get_rays / ray_rows
    and the mask construction of run_scade_wild.py:805-831 (corners first; edges only when corners are off)."""
    g = torch.Generator().manual_seed(400 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for it in range(6):
        Hh, Ww, K = ri(2, 90), ri(2, 120), ri(1, 9)
        if it == 0:
            Hh, Ww = 15, 33                                   # bands wider than the image
        intr = torch.tensor([300.0 + 400 * float(torch.rand(1, generator=g)), 300.0 + 400 * float(torch.rand(1, generator=g)),
                             Ww / 2 + float(torch.randn(1, generator=g)), Hh / 2 + float(torch.randn(1, generator=g))])
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        c2w = torch.cat([q, torch.randn(3, 1, generator=g)], -1).contiguous()
        n = ri(1, 200)
        sel = torch.stack([torch.randint(0, Hh, (n,), generator=g), torch.randint(0, Ww, (n,), generator=g)], -1)
        sel[:min(n, 4)] = torch.tensor([[0, 0], [0, Ww - 1], [Hh - 1, 0], [Hh - 1, Ww - 1]])[:min(n, 4)]
        image = torch.rand(Hh, Ww, 3, generator=g)
        hyps = torch.rand(K, Hh, Ww, generator=g) * 4.9 + 0.1
        ro, rd = O.get_rays(Hh, Ww, intr, c2w, coords=sel.float())
        want_rows = O.ray_rows(ro, rd, 0.2, 4.5)
        for corners, edges in ((False, False), (True, False), (False, True), (True, True)):
            rays, ts, th, mask = S.get_ray_batch(Hh, Ww, intr.to(dev), c2w.to(dev), sel.to(dev), 0.2, 4.5, image=image.to(dev),
                                                hypotheses=hyps.to(dev), mask_corners=corners, mask_edges=edges)
            assert_close(rays, want_rows, rtol=2e-6, atol=1e-6, what=f"rows {Hh}x{Ww}")
            assert torch.equal(ts.cpu(), image[sel[:, 0], sel[:, 1]])
            assert torch.equal(th.cpu()[..., 0], hyps[:, sel[:, 0], sel[:, 1]])
            if not (corners or edges):
                assert mask is None
                continue
            m = torch.ones(Hh, Ww)
            if corners:                                       # (:805-816)
                m[:20, :20] = 0; m[:20, -20:] = 0; m[-20:, :20] = 0; m[-20:, -20:] = 0
            else:                                             # (:818-829, an elif)
                m[:10, :] = 0; m[-10:, :] = 0; m[:, -10:] = 0; m[:, :10] = 0
            assert torch.equal(mask.cpu(), m[sel[:, 0], sel[:, 1]]), (Hh, Ww, corners, edges)
        full = ops.gen_rays(Hh, Ww, intr.to(dev), c2w.to(dev), near=0.2, far=4.5)["rays"]
        fo, fd = O.get_rays(Hh, Ww, intr, c2w)
        assert_close(full, O.ray_rows(fo, fd, 0.2, 4.5), rtol=2e-6, atol=1e-6, what="full image rows")


def test_small_exports_raw2depth_batchify_get_ray_dirs_embedder(dev):
    """The exported names no other test calls: raw2depth (run_scade_scannet.py:524-528) against the oracle's weights,
    batchify (:39-46) chunk by chunk, get_ray_dirs (helpers:285-299) against the oracle's ray directions, get_embedder's
    identity form (i = -1) and widths, DenseLayer's initialisation (xavier with the activation's gain, zero bias)."""
    g = torch.Generator().manual_seed(77)
    raw, z, d = torch.randn(9, 70, 4, generator=g), torch.sort(torch.rand(9, 70, generator=g) * 4 + 0.1, -1)[0], torch.randn(9, 3, generator=g)
    depth, std = S.raw2depth(raw.to(dev), z.to(dev), d.to(dev))
    w = O.compute_weights(raw, z, d)
    wd = torch.sum(w * z, -1)
    assert_close(depth, wd, rtol=1e-5, atol=1e-6, what="raw2depth depth")
    assert_close(std, (((z - wd.unsqueeze(-1)).pow(2) * w).sum(-1)).sqrt(), rtol=1e-4, atol=1e-5, what="raw2depth std")
    x = torch.randn(23, 5, generator=g).to(dev)
    fn = lambda t: t * 2 + 1
    assert torch.equal(S.batchify(fn, 4)(x), fn(x)) and S.batchify(fn, None) is fn
    Hh, Ww = 7, 11
    intr = torch.tensor([20.0, 21.0, 5.3, 3.1])
    c2w = torch.tensor([[0.8, 0.0, 0.6, 0.1], [0.0, 1.0, 0.0, 0.2], [-0.6, 0.0, 0.8, 0.3], [0.0, 0.0, 0.0, 1.0]])
    _, rd = O.get_rays(Hh, Ww, intr, c2w)
    assert_close(S.get_ray_dirs(Hh, Ww, intr.to(dev), c2w.to(dev)), rd, rtol=2e-6, atol=1e-7, what="get_ray_dirs")
    sel = torch.tensor([[0, 0], [6, 10], [3, 4]])
    assert_close(S.get_ray_dirs(Hh, Ww, intr.to(dev), c2w.to(dev), coords=sel.float().to(dev)), rd[sel[:, 0], sel[:, 1]],
                 rtol=2e-6, atol=1e-7, what="get_ray_dirs coords")
    ident, dim = S.get_embedder(9, -1)
    assert dim == 3 and torch.equal(ident(x), x)
    assert S.get_embedder(9, 0)[1] == 57 and S.get_embedder(0, 0)[1] == 3 and S.get_embedder(4, 0)[1] == 27
    torch.manual_seed(5)
    lin, rel = S.DenseLayer(256, 256, activation="linear"), S.DenseLayer(256, 256, activation="relu")
    assert float(lin.bias.detach().abs().max()) == 0.0 and float(rel.bias.detach().abs().max()) == 0.0
    bound = lambda gain: gain * (6.0 / 512) ** 0.5          # xavier_uniform: gain * sqrt(6 / (fan_in + fan_out))
    assert float(lin.weight.detach().abs().max()) <= bound(1.0) < float(rel.weight.detach().abs().max()) <= bound(2.0 ** 0.5)
