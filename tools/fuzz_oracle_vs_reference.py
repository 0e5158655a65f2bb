#!/usr/bin/env python3
"""Randomized check of the CPU oracle against the REAL reference (imported from /root/reference: this container only;
nothing of it travels - the summary written by --out is counts and shapes).  tools/make_golden.py pins the oracle bit for
bit on fixed shapes; this walks random ones: compositing with and without density noise, the inverse-cdf sampler (det /
given draws / numpy streams / joint), every variant of the space-carving loss with both hypothesis layouts and both
norms, the stratified jitter, ray generation, the training-batch assembly of BOTH scripts with their mask flags, and
render_rays (deterministic and jittered, lindisp, sample counts) with the three-term loss and its gradients; and the
GPU-free pieces of the host mirror: the learning-rate staircase, checkpoints in both directions.

  python tools/fuzz_oracle_vs_reference.py --seeds 40 [--out profiles/rNN_oracle_vs_reference.json]
"""
import argparse
import json
import os
import sys
import types

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG                                  # noqa: E402  (the import stubs + reference modules)

import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402

H, R, O = MG.H, MG.R, MG.O
import run_scade_wild as RW                               # noqa: E402  (reference, second script)

COUNT = {}


def same(a, b, what, group):
    a, b = a.detach(), b.detach()
    ok = a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)) \
        and torch.equal(torch.isnan(a), torch.isnan(b))
    c = COUNT.setdefault(group, [0, 0])
    c[0] += 1
    if not ok:
        c[1] += 1
        err = (a.double() - b.double()).abs().max().item() if a.shape == b.shape else float("nan")
        print(f"MISMATCH {group}: {what}: shapes {tuple(a.shape)} / {tuple(b.shape)}, max abs diff {err:g}", flush=True)
    return ok


def one(seed):
    g = torch.Generator().manual_seed(5000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    rnd = lambda *s: torch.rand(*s, generator=g)
    rnn = lambda *s: torch.randn(*s, generator=g)
    N = ri(1, 40)
    # ---- compositing (density noise: raw_noise_std > 0 draws torch.randn inside the reference)
    S = ri(2, 200)
    raw = rnn(N, S, 4)
    z = torch.sort(rnd(N, S) * 4.9 + 0.1, -1)[0]
    d = rnn(N, 3) * 1.5
    r1 = raw.clone().requires_grad_(True)
    outs = R.raw2outputs(r1, z, d, 0, pytest=False)
    G = [rnn(*o.shape) for o in outs]
    sum((o * gg).sum() for o, gg in zip(outs, G)).backward()
    r2 = raw.clone().requires_grad_(True)
    oo = O.raw2outputs(r2, z, d)
    sum((o * gg).sum() for o, gg in zip(oo, G)).backward()
    for a, b, n in zip(oo, outs, ["rgb", "disp", "acc", "w", "depth"]):
        same(a, b, f"N={N} S={S} {n}", "raw2outputs")
    same(r2.grad, r1.grad, f"N={N} S={S} grad", "raw2outputs")
    std = 0.5
    torch.manual_seed(seed)
    on = R.raw2outputs(raw, z, d, std, pytest=False)
    torch.manual_seed(seed)
    noise = torch.randn(raw[..., 3].shape) * std
    for a, b, n in zip(O.raw2outputs(raw, z, d, noise), on, ["rgb", "disp", "acc", "w", "depth"]):
        same(a, b, f"noise N={N} S={S} {n}", "raw2outputs")
    # ---- sampler
    M, Sn = ri(2, 200), ri(1, 150)
    bins = torch.sort(rnd(N, M) * 4.9 + 0.1, -1)[0]
    w = rnd(N, M - 1) ** 4
    if N > 1:
        w[0] = 0.0
    w1 = w.clone().requires_grad_(True)
    u = rnd(N, Sn)
    s_u, _ = H.sample_pdf_return_u(bins, w1, Sn, det=False, load_u=u)
    Gs = rnn(N, Sn)
    (s_u * Gs).sum().backward()
    w2 = w.clone().requires_grad_(True)
    s_o = O.sample_pdf(bins, w2, u)
    (s_o * Gs).sum().backward()
    same(s_o, s_u, f"N={N} M={M} S={Sn} load_u", "sample_pdf")
    same(w2.grad, w1.grad, f"N={N} M={M} S={Sn} grad", "sample_pdf")
    same(O.sample_pdf(bins, w, O.draw_u(N, Sn, det=True)), H.sample_pdf(bins, w, Sn, det=True), "det", "sample_pdf")
    same(O.sample_pdf(bins, w, O.draw_u(N, Sn, det=False, pytest=True)), H.sample_pdf(bins, w, Sn, det=False, pytest=True),
         "pytest", "sample_pdf")
    torch.manual_seed(seed)
    sj = H.sample_pdf_joint(bins, w, Sn, det=False)
    torch.manual_seed(seed)
    uj = torch.rand(Sn)
    same(O.sample_pdf(bins, w, uj.expand(N, Sn)), sj, "joint", "sample_pdf")
    # ---- space carving
    K, P = ri(1, 50), ri(1, 150)
    pred = rnd(N, P) * 5
    mask = (rnd(N) > 0.3).float()
    for per_sample in (False, True):
        hyp = rnd(K, N, P if per_sample else 1) * 4.9 + 0.1
        for kw in (dict(), dict(is_joint=True), dict(mask=mask), dict(threshold=0.3), dict(norm_p=1),
                   dict(is_joint=True, mask=mask, threshold=0.3), dict(mask=mask, threshold=0.05, norm_p=1)):
            p1, h1 = pred.clone().requires_grad_(True), hyp.clone().requires_grad_(True)
            l1 = H.compute_space_carving_loss(p1, h1, **kw)
            l1.backward()
            p2, h2 = pred.clone().requires_grad_(True), hyp.clone().requires_grad_(True)
            l2 = O.compute_space_carving_loss(p2, h2, **kw)
            l2.backward()
            tag = f"N={N} P={P} K={K} per_sample={per_sample} {sorted(kw)}"
            same(l2, l1, tag, "space_carving")
            same(p2.grad, p1.grad, tag + " d pred", "space_carving")
            same(h2.grad, h1.grad, tag + " d hyp", "space_carving")
    # ---- jitter, rays, losses
    zz = torch.sort(rnd(N, S) * 4.9 + 0.1, -1)[0]
    zp = R.perturb_z_vals(zz, True)
    np.random.seed(0)
    same(O.perturb_z_vals(zz, torch.Tensor(np.random.rand(N, S))), zp, f"N={N} S={S}", "perturb_z_vals")
    Hh, Ww = ri(2, 60), ri(2, 80)
    intr = torch.tensor([300.0 + 300 * float(rnd(1)), 300.0 + 300 * float(rnd(1)), Ww / 2 + float(rnn(1)), Hh / 2 + float(rnn(1))])
    q, _ = torch.linalg.qr(rnn(3, 3))
    c2w = torch.cat([torch.cat([q, rnn(3, 1)], -1), torch.tensor([[0., 0., 0., 1.]])], 0)
    ro, rd = H.get_rays(Hh, Ww, intr, c2w)
    oro, ord_ = O.get_rays(Hh, Ww, intr, c2w)
    same(ord_, rd, f"{Hh}x{Ww} d", "get_rays")
    same(oro.expand(rd.shape), ro, f"{Hh}x{Ww} o", "get_rays")
    x, y = rnd(N, 3), rnd(N, 3)
    same(O.img2mse(x, y), H.img2mse(x, y), "mse", "img2mse")
    # ---- the training-batch assembly of both scripts: gathers + mask flags (corner mask first, edge mask an elif)
    V, Kh = 2, ri(1, 6)
    images, depths, valid = rnd(V, Hh, Ww, 3), rnd(V, Hh, Ww, 1), (rnd(V, Hh, Ww, 1) > 0.5)
    poses = torch.stack([c2w, c2w], 0)
    intrs = torch.stack([intr, intr], 0)
    hyps = rnd(V, Kh, Hh, Ww, 1) * 4.9 + 0.1
    n_rand = ri(1, min(64, Hh * Ww))
    for mod, flags in ((R, [(False, False), (True, False)]), (RW, [(False, False), (True, False), (False, True), (True, True)])):
        for corners, edges in flags:
            args = types.SimpleNamespace(N_rand=n_rand, mask_corners=corners, mask_edges=edges)
            np.random.seed(seed)
            out = mod.get_ray_batch_from_one_image_hypothesis_idx(Hh, Ww, 1, images, depths, valid, poses, intrs, hyps, args, None, None)
            batch_rays, target_s, _, _, _, target_h, sc_mask, _ = out
            np.random.seed(seed)
            sel = torch.as_tensor(np.random.choice(Hh * Ww, size=[n_rand], replace=False))
            rows, cols = sel // Ww, sel % Ww
            o2, d2 = O.get_rays(Hh, Ww, intr, c2w, coords=torch.stack([rows, cols], -1).float())
            tag = f"{mod.__name__} {Hh}x{Ww} corners={corners} edges={edges}"
            same(d2, batch_rays[1], tag + " rays_d", "batch_assembly")
            same(o2, batch_rays[0], tag + " rays_o", "batch_assembly")
            same(images[1][rows, cols], target_s, tag + " target_s", "batch_assembly")
            same(hyps[1][:, rows, cols], target_h, tag + " target_h", "batch_assembly")
            m = torch.ones(Hh, Ww)
            if corners:
                m[:20, :20] = 0; m[:20, -20:] = 0; m[-20:, :20] = 0; m[-20:, -20:] = 0
            elif edges and mod is RW:
                m[:10, :] = 0; m[-10:, :] = 0; m[:, -10:] = 0; m[:, :10] = 0
            if corners or (edges and mod is RW):
                same(m[rows, cols], sc_mask, tag + " mask", "batch_assembly")
            else:
                c = COUNT.setdefault("batch_assembly", [0, 0])
                c[0] += 1
                if sc_mask is not None:
                    c[1] += 1
                    print("MISMATCH batch_assembly: mask expected None", tag)
    return N


def render(seed):
    """render_rays + the three-term loss + every gradient (slow: a few seeds)"""
    g = torch.Generator().manual_seed(6000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N, ns, ni, K = ri(1, 12), ri(3, 80), ri(1, 100), ri(1, 12)
    lindisp, jitter = bool(ri(0, 1)), bool(ri(0, 1))
    pc, pf = O.nerf_init(100 + seed), O.nerf_init(200 + seed)
    coarse, fine = MG.ref_nerf(pc), MG.ref_nerf(pf)
    embed_fn, _ = H.get_embedder(9, 0)
    embeddirs_fn, _ = H.get_embedder(0, 0)
    bbc, bbs = torch.randn(3, generator=g) * 0.1, torch.tensor(0.2)

    def query(pts, vd, cam, fn):
        return R.run_network(pts, vd, cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, bb_center=bbc, bb_scale=bbs,
                             netchunk=1024 * 64)
    rays = O.synthetic_rays(N, seed=300 + seed, unit_dirs=False)
    tgt = torch.rand(N, 3, generator=g)
    hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
    ret = R.render_rays(rays, True, coarse, query, ns, embedded_cam=torch.tensor(()), N_importance=ni, network_fine=fine,
                        perturb=1.0 if jitter else 0.0, lindisp=lindisp, retraw=True, pytest=jitter)
    loss = H.img2mse(ret["rgb_map"], tgt) + 0.007 * H.compute_space_carving_loss(ret["pred_hyp"], hyp) + H.img2mse(ret["rgb0"], tgt)
    loss.backward()
    kw = {}
    if jitter:
        np.random.seed(0)
        kw["t_rand"] = torch.Tensor(np.random.rand(N, ns))
        np.random.seed(0)
        kw["u_coarse"] = kw["u_fine"] = torch.Tensor(np.random.rand(N, ni))
    po_c = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    po_f = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
    ro = O.render_rays(rays, po_c, po_f, bbc, bbs, n_samples=ns, n_importance=ni, lindisp=lindisp, retraw=True, **kw)
    tag = f"N={N} ns={ns} ni={ni} lindisp={lindisp} jitter={jitter}"
    for k in ret:
        same(ro[k], ret[k], tag + " ." + k, "render_rays")
    lo = O.train_loss(ro, tgt, hyp)[0]
    same(lo, loss, tag + " loss", "render_rays")
    lo.backward()
    for net, po in ((coarse, po_c), (fine, po_f)):
        for k, p in net.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            go = po[k].grad if po[k].grad is not None else torch.zeros_like(p)
            same(go, gr, tag + " d/d " + k, "render_rays gradients")


def render_joint(seed):
    """render_rays(is_joint=True, perturb=1) on torch's own random stream: the jitter [N,S], the coarse sampler's draws
    [N,Si], then ONE row [Si] shared by all rays for the depth-hypothesis sampler (helpers:498-503) - the oracle gets the
    same three draws handed in"""
    g = torch.Generator().manual_seed(8000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N, ns, ni = ri(1, 10), ri(3, 50), ri(1, 60)
    pc, pf = O.nerf_init(300 + seed), O.nerf_init(400 + seed)
    coarse, fine = MG.ref_nerf(pc), MG.ref_nerf(pf)
    embed_fn, _ = H.get_embedder(9, 0)
    embeddirs_fn, _ = H.get_embedder(0, 0)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)

    def query(pts, vd, cam, fn):
        return R.run_network(pts, vd, cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, bb_center=bbc, bb_scale=bbs,
                             netchunk=1024 * 64)
    rays = O.synthetic_rays(N, seed=500 + seed)
    with torch.no_grad():
        torch.manual_seed(seed)
        ret = R.render_rays(rays, True, coarse, query, ns, embedded_cam=torch.tensor(()), N_importance=ni, network_fine=fine,
                            perturb=1.0, is_joint=True)
        torch.manual_seed(seed)
        t_rand, uc, uj = torch.rand(N, ns), torch.rand(N, ni), torch.rand(ni)
        ro = O.render_rays(rays, pc, pf, bbc, bbs, n_samples=ns, n_importance=ni, t_rand=t_rand, u_coarse=uc,
                           u_fine=uj.unsqueeze(0).repeat(N, 1))
    for k in ret:
        same(ro[k], ret[k], f"N={N} ns={ns} ni={ni} .{k}", "render_rays is_joint")


def host_mirror():
    """the pieces of the host-side mirror that run without a GPU, against the reference's own: the learning-rate staircase
    and checkpoints in both directions (ours into the reference's DataParallel-wrapped modules, a reference one into ours)"""
    import random
    import tempfile
    from train_utils.hyperparameter_update import get_learning_rate
    from scade_amd import scene
    from scade_amd.parallel import staircase_lr
    from scade_amd.run_nerf_helpers import NeRF
    random.seed(0)
    c = COUNT.setdefault("learning-rate staircase", [0, 0])
    for _ in range(20000):
        lr0 = random.choice([5e-4, 1e-3, 1e-7, random.random() * 1e-2])
        rate = random.choice([0.1, 0.5, 0.9, random.random()])
        step = random.choice([1, 4, 400000, 250000, random.randint(1, 10 ** 6)])
        it = random.randint(1, 2 * 10 ** 6)
        c[0] += 1
        c[1] += get_learning_rate(lr0, it, step, rate, staircase=True) != staircase_lr(lr0, rate, step, it)
    mk = lambda: NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, input_ch_cam=0, use_viewdirs=True)
    ours_c, ours_f = mk(), mk()
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "exp"))
    scene.save_checkpoint(os.path.join(d, "exp", "001000.tar"), 1000, ours_c, ours_f, torch.zeros(3, 1), torch.ones(3, 1))
    ck = torch.load(os.path.join(d, "exp", "001000.tar"), weights_only=False)
    ref_c = torch.nn.DataParallel(MG.ref_nerf(O.nerf_init(1)))       # (create_nerf wraps both networks: :422, :431)
    ref_f = torch.nn.DataParallel(MG.ref_nerf(O.nerf_init(2)))
    ref_c.load_state_dict(ck["network_fn_state_dict"])
    ref_f.load_state_dict(ck["network_fine_state_dict"])
    for (k, a), b in zip(ref_c.module.state_dict().items(), ours_c.state_dict().values()):
        same(a, b, "ours -> reference " + k, "checkpoint interop")
    c = COUNT["checkpoint interop"]
    c[0] += 1
    c[1] += sorted(ck) != ["depth_scales", "depth_shifts", "global_step", "network_fine_state_dict", "network_fn_state_dict",
                           "optimizer_state_dict"]
    ref_c, ref_f = torch.nn.DataParallel(MG.ref_nerf(O.nerf_init(3))), torch.nn.DataParallel(MG.ref_nerf(O.nerf_init(4)))
    torch.save({"global_step": 7000, "network_fn_state_dict": ref_c.state_dict(), "network_fine_state_dict": ref_f.state_dict(),
                "optimizer_state_dict": {}, "depth_shifts": torch.zeros(3, 1), "depth_scales": torch.ones(3, 1)},
               os.path.join(d, "exp", "007000.tar"))
    ck = scene.load_checkpoint(d, "exp")
    ours_c.load_reference_state_dict(ck["network_fn_state_dict"])
    ours_f.load_reference_state_dict(ck["network_fine_state_dict"])
    for net, ref in ((ours_c, ref_c), (ours_f, ref_f)):
        for (k, a), b in zip(ref.module.state_dict().items(), net.state_dict().values()):
            same(b, a, "reference -> ours " + k, "checkpoint interop")
    c[0] += 1
    c[1] += ck["global_step"] != 7000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--render-seeds", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(8)
    for s in range(a.seeds):
        one(s)
    for s in range(a.render_seeds):
        render(s)
        render_joint(s)
    host_mirror()
    bad = sum(v[1] for v in COUNT.values())
    for k, (n, b) in COUNT.items():
        print(f"{k:24s} {n:6d} comparisons, {b} mismatches")
    print("oracle == reference bit for bit" if not bad else f"{bad} MISMATCHES")
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"command": f"python tools/fuzz_oracle_vs_reference.py --seeds {a.seeds} --render-seeds {a.render_seeds}   (build container, CPU; "
                                  "the reference is imported from /root/reference and does not travel)",
                       "what": "random shapes and flags: every oracle function against the reference's own, bit for bit",
                       "torch": torch.__version__, "comparisons": {k: v[0] for k, v in COUNT.items()},
                       "mismatches": {k: v[1] for k, v in COUNT.items()}}, f, indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
