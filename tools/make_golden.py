#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, survey/build container only) and, on the way, assert that the
CPU oracle (oracle/scade_oracle.py) reproduces it.

The reference's Python never travels: only inputs + expected outputs (data) are
written.  Run:  python tools/make_golden.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

_STUB = {"cv2", "configargparse", "skimage", "lpips", "torchvision", "imageio", "tensorboard"}


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Stub(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return None


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _STUB or name == "torch.utils.tensorboard":
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, m):
        pass


sys.meta_path.insert(0, _Finder())

import numpy as np  # noqa: E402
import torch  # noqa: E402

import model.run_nerf_helpers as H  # noqa: E402  (reference)
import run_scade_scannet as R  # noqa: E402  (reference)
from oracle import scade_oracle as O  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def same(a, b, what):
    a, b = a.detach(), b.detach()
    if not (a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))
            and torch.equal(torch.isnan(a), torch.isnan(b))):
        err = (a.double() - b.double()).abs().max().item()
        raise SystemExit(f"ORACLE != REFERENCE for {what}: max abs diff {err:g}")
    print(f"  oracle == reference (bit-exact): {what}")


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (np_(v) if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_nerf(params):
    net = H.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3,
                 input_ch_cam=0, use_viewdirs=True)
    net.load_state_dict(params)
    return net


def weight_digest(params):
    """Small fingerprint so the GPU box can check nerf_init(seed) reproduces the
    same weights (RNG drift guard)."""
    d = {}
    for k, v in params.items():
        d[k] = np.array([v.double().sum().item(), v.double().abs().sum().item(),
                         float(v.flatten()[0]), float(v.flatten()[-1])])
    return d


SUB = 97  # stride of the sub-sample kept for the big gradient tensors


def grad_subsample(g):
    f = g.flatten()
    return f if f.numel() <= 4096 else f[::SUB]


# ---------------------------------------------------------------- F1 embed
def f1_embed():
    torch.manual_seed(0)
    x = torch.rand(64, 3) * 2 - 1
    x[0] = torch.tensor([0.0, 1.0, -1.0])
    x[1] = torch.tensor([0.5, -0.25, 0.125])
    fn, dim = H.get_embedder(9, 0)
    y = fn(x)
    assert dim == 57
    same(O.embed(x, 9), y, "embed(9)")
    fn0, dim0 = H.get_embedder(0, 0)
    same(O.embed(x, 0), fn0(x), "embed(0)")
    assert dim0 == 3
    save("f1_embed", x=x, y=y)


# ---------------------------------------------------------------- F2 mlp
def f2_mlp():
    seed = 1234
    params = O.nerf_init(seed)
    # non-zero biases so the bias path is pinned too
    g = torch.Generator().manual_seed(99)
    for k in params:
        if k.endswith(".bias"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    net = ref_nerf(params)
    torch.manual_seed(1)
    pts = torch.rand(256, 3) * 2 - 1
    vd = torch.randn(256, 3)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    x = torch.cat([O.embed(pts, 9), vd], -1)
    G = torch.randn(256, 4)
    out = net(x)
    (out * G).sum().backward()
    ref_grads = {k: p.grad.clone() for k, p in net.named_parameters()}

    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out_o = O.nerf_forward(po, x)
    same(out_o, out, "NeRF.forward")
    (out_o * G).sum().backward()
    for k in po:
        same(po[k].grad, ref_grads[k], f"dNeRF/d{k}")

    arrs = dict(seed=seed, bias_seed=99, pts=pts, viewdirs=vd, x=x, G=G, out=out)
    for k, v in weight_digest(params).items():
        arrs["digest/" + k] = v
    for k, v in ref_grads.items():
        arrs["grad/" + k] = grad_subsample(v)
    save("f2_mlp", **arrs)


# ---------------------------------------------------------------- F3 composite
def f3_composite():
    for S in (64, 192):
        torch.manual_seed(10 + S)
        N = 32
        raw = torch.randn(N, S, 4)
        raw[..., 3] = torch.nn.functional.softplus(raw[..., 3] * 3.0, beta=10)
        raw[3, :, 3] = 0.0                     # empty ray
        raw[4, 5, 3] = 1e4                     # opaque wall: 1-alpha == 0 downstream
        z = torch.sort(torch.rand(N, S) * 4.9 + 0.1, -1)[0]
        d = torch.randn(N, 3) * 1.5
        raw.requires_grad_(True)
        outs = R.raw2outputs(raw, z, d, 0, pytest=False)
        G = [torch.randn_like(o) for o in outs]
        sum((o * g).sum() for o, g in zip(outs, G)).backward()
        graw = raw.grad.clone()
        raw2 = raw.detach().clone().requires_grad_(True)
        outs_o = O.raw2outputs(raw2, z, d)
        for a, b, n in zip(outs_o, outs, ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]):
            same(a, b, f"raw2outputs[{S}].{n}")
        sum((o * g).sum() for o, g in zip(outs_o, G)).backward()
        same(raw2.grad, graw, f"d raw2outputs[{S}]/d raw")
        same(O.compute_weights(raw.detach(), z, d), R.compute_weights(raw.detach(), z, d),
             f"compute_weights[{S}]")
        save(f"f3_composite_{S}", raw=raw, z=z, d=d, rgb_map=outs[0], disp_map=outs[1],
             acc_map=outs[2], weights=outs[3], depth_map=outs[4],
             G_rgb=G[0], G_disp=G[1], G_acc=G[2], G_w=G[3], G_depth=G[4], grad_raw=graw)


# ---------------------------------------------------------------- F4 sample_pdf
def f4_sample_pdf():
    for M in (63, 191):
        torch.manual_seed(20 + M)
        N, S = 32, 128
        bins = torch.sort(torch.rand(N, M) * 4.9 + 0.1, -1)[0]
        w = torch.rand(N, M - 1) ** 4
        w[0] = 0.0                               # all-zero weights
        w[1] = 0.0
        w[1, 7] = 1.0                            # one-hot
        w[2, : (M - 1) // 2] = 0.0               # empty front half
        w.requires_grad_(True)
        # det (linspace u)
        s_det = H.sample_pdf(bins, w, S, det=True)
        same(O.sample_pdf(bins, w, O.draw_u(N, S, det=True)), s_det, f"sample_pdf[{M}] det")
        # explicit u through load_u
        u = torch.rand(N, S)
        u[5, 0] = 0.0
        u[5, 1] = 1.0
        s_u, u_back = H.sample_pdf_return_u(bins, w, S, det=False, load_u=u)
        assert torch.equal(u_back, u)
        G = torch.randn(N, S)
        (s_u * G).sum().backward()
        gw = w.grad.clone()
        cdf = O.pdf_to_cdf(w.detach())
        inds = torch.searchsorted(cdf, u, right=True)
        w2 = w.detach().clone().requires_grad_(True)
        s_o = O.sample_pdf(bins, w2, u)
        same(s_o, s_u, f"sample_pdf_return_u[{M}] load_u")
        (s_o * G).sum().backward()
        same(w2.grad, gw, f"d sample_pdf[{M}]/d w")
        same(O.invert_cdf(bins, cdf, u)[1], inds, f"inds[{M}]")
        # pytest=True numpy streams
        s_py = H.sample_pdf(bins, w.detach(), S, det=False, pytest=True)
        same(O.sample_pdf(bins, w.detach(), O.draw_u(N, S, det=False, pytest=True)), s_py,
             f"sample_pdf[{M}] pytest")
        # joint: one u[S] shared by all rays
        torch.manual_seed(77)
        s_joint = H.sample_pdf_joint(bins, w.detach(), S, det=False)
        torch.manual_seed(77)
        u_joint = torch.rand(S)
        same(O.sample_pdf(bins, w.detach(), u_joint.expand(N, S)), s_joint, f"sample_pdf_joint[{M}]")
        save(f"f4_sample_pdf_{M}", bins=bins, w=w, u=u, cdf=cdf, inds=inds, samples_u=s_u,
             samples_det=s_det, samples_pytest=s_py, u_joint=u_joint, samples_joint=s_joint,
             G=G, grad_w=gw)


# ---------------------------------------------------------------- F5 carve
def f5_carve():
    for K in (20, 40):
        torch.manual_seed(30 + K)
        N, P = 32, 128
        pred = (torch.rand(N, P) * 4.9 + 0.1)
        hyp = (torch.rand(K, N, 1) * 4.9 + 0.1)
        hyp[0, 3, 0] = 1.0
        hyp[1, 3, 0] = 3.0
        pred[3, :4] = 2.0                        # exact tie between hyp 0 and 1 ...
        hyp[2:, 3, 0] = 10.0                     # ... all other hyps far away
        pred[4, 0] = hyp[5, 4, 0]                # exact zero distance
        mask = (torch.rand(N) > 0.3).float()
        arrs = dict(pred=pred, hyp=hyp, mask=mask)
        variants = {
            "default": dict(),
            "mask": dict(mask=mask),
            "thr": dict(threshold=0.05),
            "joint": dict(is_joint=True),
            "p1": dict(norm_p=1),
            "mask_thr": dict(mask=mask, threshold=0.05),
        }
        for name, kw in variants.items():
            p = pred.clone().requires_grad_(True)
            h = hyp.clone().requires_grad_(True)
            loss = H.compute_space_carving_loss(p, h, **kw)
            loss.backward()
            p2 = pred.clone().requires_grad_(True)
            h2 = hyp.clone().requires_grad_(True)
            lo = O.compute_space_carving_loss(p2, h2, **kw)
            lo.backward()
            same(lo, loss, f"carve[{K}].{name}")
            same(p2.grad, p.grad, f"d carve[{K}].{name}/d pred")
            same(h2.grad, h.grad, f"d carve[{K}].{name}/d hyp")
            arrs[f"{name}/loss"] = loss
            arrs[f"{name}/grad_pred"] = p.grad
            arrs[f"{name}/grad_hyp"] = h.grad
        save(f"f5_carve_{K}", **arrs)


# ---------------------------------------------------------------- F5b carve, cached-quantile hypotheses
def f5_carve_knp():
    """target_hypothesis [K,N,P]: "each quantile here already picked a hypothesis" (helpers:100-102)."""
    torch.manual_seed(95)
    K, N, P = 20, 16, 128
    pred = (torch.rand(N, P) * 4.9 + 0.1)
    hyp = (torch.rand(K, N, P) * 4.9 + 0.1)
    hyp[0, 3, :4] = 1.0
    hyp[1, 3, :4] = 3.0
    pred[3, :4] = 2.0                        # exact tie between hyp 0 and 1 ...
    hyp[2:, 3, :4] = 10.0                    # ... all other hyps far away
    pred[4, 0] = hyp[5, 4, 0]                # exact zero distance
    mask = (torch.rand(N) > 0.3).float()
    arrs = dict(pred=pred, hyp=hyp, mask=mask)
    variants = {"default": dict(), "mask": dict(mask=mask), "thr": dict(threshold=0.05),
                "joint": dict(is_joint=True), "joint_mask_thr": dict(is_joint=True, mask=mask, threshold=0.05),
                "mask_thr": dict(mask=mask, threshold=0.05)}
    for name, kw in variants.items():
        p = pred.clone().requires_grad_(True)
        h = hyp.clone().requires_grad_(True)
        loss = H.compute_space_carving_loss(p, h, **kw)
        loss.backward()
        p2 = pred.clone().requires_grad_(True)
        h2 = hyp.clone().requires_grad_(True)
        lo = O.compute_space_carving_loss(p2, h2, **kw)
        lo.backward()
        same(lo, loss, f"carve_knp.{name}")
        same(p2.grad, p.grad, f"d carve_knp.{name}/d pred")
        same(h2.grad, h.grad, f"d carve_knp.{name}/d hyp")
        arrs[f"{name}/loss"] = loss
        arrs[f"{name}/grad_pred"] = p.grad
        arrs[f"{name}/grad_hyp"] = h.grad
    save("f5_carve_knp", **arrs)


# ---------------------------------------------------------------- F7 perturb
def f7_perturb():
    torch.manual_seed(40)
    near, far = 0.1, 5.0
    t = torch.linspace(0.0, 1.0, steps=64)
    z = (near * (1.0 - t) + far * t).expand(32, 64).contiguous()
    zp = R.perturb_z_vals(z, True)               # pytest=True -> numpy seeded t_rand
    np.random.seed(0)
    t_rand = torch.Tensor(np.random.rand(32, 64))
    same(O.perturb_z_vals(z, t_rand), zp, "perturb_z_vals")
    save("f7_perturb", z=z, t_rand=t_rand, out=zp)


# ---------------------------------------------------------------- F6 render_rays
def f6_render():
    N, K = 32, 20
    seed_c, seed_f = 11, 12
    pc, pf = O.nerf_init(seed_c), O.nerf_init(seed_f)
    g = torch.Generator().manual_seed(5)
    for p in (pc, pf):                           # non-zero biases
        for k in p:
            if k.endswith(".bias"):
                p[k] = 0.05 * torch.randn(p[k].shape, generator=g)
    coarse, fine = ref_nerf(pc), ref_nerf(pf)
    embed_fn, _ = H.get_embedder(9, 0)
    embeddirs_fn, _ = H.get_embedder(0, 0)
    bb_center, bb_scale = torch.zeros(3), torch.tensor(0.2)

    def query(pts, vd, cam, fn):
        return R.run_network(pts, vd, cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                             bb_center=bb_center, bb_scale=bb_scale, netchunk=1024 * 64)

    torch.manual_seed(6)
    o = 0.1 * torch.randn(N, 3)
    d = torch.randn(N, 3) * 1.3                  # non-unit directions
    vd = d / d.norm(dim=-1, keepdim=True)
    rays = torch.cat([o, d, torch.full((N, 1), 0.1), torch.full((N, 1), 5.0), vd], -1)
    target_s = torch.rand(N, 3)
    hyp0 = torch.rand(K, N, 1) * 4.9 + 0.1
    scale = torch.ones(1, requires_grad=True)
    shift = torch.zeros(1, requires_grad=True)

    # --- det path (test render)
    with torch.no_grad():
        ret = R.render_rays(rays, True, coarse, query, 64, embedded_cam=torch.tensor(()),
                            N_importance=128, network_fine=fine, perturb=0.0, retraw=True)
        ret_o = O.render_rays(rays, pc, pf, bb_center, bb_scale, retraw=True)
    for k in ret:
        same(ret_o[k], ret[k], f"render_rays det .{k}")
    arrs = dict(seed_coarse=seed_c, seed_fine=seed_f, bias_seed=5, rays=rays, target_s=target_s,
                hyp=hyp0, bb_center=bb_center, bb_scale=bb_scale)
    for k, v in ret.items():
        arrs["det/" + k] = v
    for tag, p in (("coarse", pc), ("fine", pf)):
        for k, v in weight_digest(p).items():
            arrs[f"digest_{tag}/{k}"] = v

    # --- train path: perturb=1 with the numpy pytest streams
    ret = R.render_rays(rays, True, coarse, query, 64, embedded_cam=torch.tensor(()),
                        N_importance=128, network_fine=fine, perturb=1.0, retraw=True, pytest=True)
    target_h = hyp0 * scale + shift
    img_loss = H.img2mse(ret["rgb_map"], target_s)
    carve = H.compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=False, norm_p=2,
                                         threshold=0.0)
    img_loss0 = H.img2mse(ret["rgb0"], target_s)
    loss = img_loss + 0.007 * carve + img_loss0
    loss.backward()

    np.random.seed(0)
    t_rand = torch.Tensor(np.random.rand(N, 64))
    np.random.seed(0)
    u_py = torch.Tensor(np.random.rand(N, 128))
    po_c = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    po_f = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
    sc2 = torch.ones(1, requires_grad=True)
    sh2 = torch.zeros(1, requires_grad=True)
    ret_o = O.render_rays(rays, po_c, po_f, bb_center, bb_scale, t_rand=t_rand, u_coarse=u_py,
                          u_fine=u_py, retraw=True)
    for k in ret:
        same(ret_o[k], ret[k], f"render_rays train .{k}")
    lo, il, cv, il0 = O.train_loss(ret_o, target_s, hyp0 * sc2 + sh2)
    same(lo, loss, "train loss")
    lo.backward()
    for k, p in coarse.named_parameters():
        gref = p.grad if p.grad is not None else torch.zeros_like(p)
        go = po_c[k].grad if po_c[k].grad is not None else torch.zeros_like(p)
        same(go, gref, f"d loss/d coarse.{k}")
        arrs["grad_coarse/" + k] = grad_subsample(gref)
    for k, p in fine.named_parameters():
        gref = p.grad if p.grad is not None else torch.zeros_like(p)
        go = po_f[k].grad if po_f[k].grad is not None else torch.zeros_like(p)
        same(go, gref, f"d loss/d fine.{k}")
        arrs["grad_fine/" + k] = grad_subsample(gref)
    same(sc2.grad, scale.grad, "d loss/d scale")
    same(sh2.grad, shift.grad, "d loss/d shift")
    for k, v in ret.items():
        arrs["train/" + k] = v
    arrs.update({"train/t_rand": t_rand, "train/u": u_py, "train/loss": loss,
                 "train/img_loss": img_loss, "train/carve": carve, "train/img_loss0": img_loss0,
                 "train/grad_scale": scale.grad, "train/grad_shift": shift.grad})
    save("f6_render", **arrs)


# ---------------------------------------------------------------- F8 rays
def f8_rays():
    torch.manual_seed(50)
    Hh, Ww = 24, 32
    intrinsic = torch.tensor([28.9, 29.3, 15.7, 12.2])
    ang = torch.tensor(0.3)
    c2w = torch.tensor([[torch.cos(ang), 0.0, torch.sin(ang), 0.5],
                        [0.1, 0.99, 0.05, -0.25],
                        [-torch.sin(ang), 0.02, torch.cos(ang), 1.5],
                        [0.0, 0.0, 0.0, 1.0]])
    ro, rd = H.get_rays(Hh, Ww, intrinsic, c2w)                       # full image
    same(O.get_rays(Hh, Ww, intrinsic, c2w)[1], rd, "get_rays full")
    coords_f = torch.stack(torch.meshgrid(torch.linspace(0, Hh - 1, Hh), torch.linspace(0, Ww - 1, Ww),
                                          indexing='ij'), -1)
    np.random.seed(3)
    sel = H.select_coordinates(coords_f, 64)                          # [64,2] long (row, col)
    rays_o = ro[sel[:, 0], sel[:, 1]]
    rays_d = rd[sel[:, 0], sel[:, 1]]
    # the row assembly of render_hyp (run_scade_scannet.py:200-219) with use_viewdirs
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    near, far = 0.1, 5.0
    rows = torch.cat([rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]),
                      far * torch.ones_like(rays_d[..., :1]), viewdirs], -1)
    same(O.ray_rows(rays_o, rays_d, near, far), rows, "ray rows")
    image = torch.rand(Hh, Ww, 3)
    hyps = torch.rand(5, Hh, Ww, 1) * 4.9 + 0.1
    target_s = image[sel[:, 0], sel[:, 1]]
    target_h = hyps[:, sel[:, 0], sel[:, 1]]                          # [K,N,1]
    save("f8_rays", H=Hh, W=Ww, intrinsic=intrinsic, c2w=c2w, rays_o_full=ro, rays_d_full=rd, sel=sel,
         rows=rows, image=image, hyps=hyps, target_s=target_s, target_h=target_h, near=near, far=far)


# ---------------------------------------------------------------- F9 render(): the plumbing around render_rays
def f9_render_image():
    """render() of the reference (run_scade_scannet.py:80-155) on an 18 x 40 image, 16 + 24 samples: full image through a
    chunk length that does not divide it, the 5.33:9 centre crop, a given ray batch, and c2w_staticcam (view directions
    of one camera on the rays of another)."""
    Hh, Ww = 18, 40
    seed_c, seed_f = 21, 22
    pc, pf = O.nerf_init(seed_c), O.nerf_init(seed_f)
    coarse, fine = ref_nerf(pc), ref_nerf(pf)
    embed_fn, _ = H.get_embedder(9, 0)
    embeddirs_fn, _ = H.get_embedder(0, 0)
    bb_center, bb_scale = torch.tensor([0.05, -0.02, 0.1]), torch.tensor(0.2)

    def query(pts, vd, cam, fn):
        return R.run_network(pts, vd, cam, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                             bb_center=bb_center, bb_scale=bb_scale, netchunk=1024 * 64)

    intrinsic = torch.tensor([35.0, 36.5, 19.6, 9.3])
    ang = torch.tensor(-0.4)
    c2w = torch.tensor([[torch.cos(ang), 0.0, torch.sin(ang), 0.2], [0.05, 0.995, 0.03, -0.1],
                        [-torch.sin(ang), 0.01, torch.cos(ang), 0.7]])
    c2w_b = torch.tensor([[1.0, 0.0, 0.0, -0.3], [0.0, 1.0, 0.0, 0.2], [0.0, 0.0, 1.0, 0.4]])
    kw = dict(near=0.1, far=5.0, use_viewdirs=True, network_fn=coarse, network_query_fn=query, N_samples=16,
              N_importance=24, network_fine=fine, perturb=0.0, embedded_cam=torch.tensor(()))
    arrs = dict(H=Hh, W=Ww, intrinsic=intrinsic, c2w=c2w, c2w_b=c2w_b, bb_center=bb_center, bb_scale=bb_scale,
                seed_coarse=seed_c, seed_fine=seed_f)
    torch.manual_seed(3)
    ro, rd = H.get_rays(Hh, Ww, intrinsic, c2w)
    sel = torch.randperm(Hh * Ww)[:53]
    batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    arrs["batch"] = batch
    with torch.no_grad():
        cases = {"full": R.render(Hh, Ww, intrinsic, chunk=37, c2w=c2w, **kw),
                 "crop": R.render(Hh, Ww, intrinsic, chunk=64, c2w=c2w, with_5_9=True, **kw),
                 "batch": R.render(Hh, Ww, intrinsic, chunk=20, rays=batch, **kw),
                 "static": R.render(Hh, Ww, intrinsic, chunk=128, c2w=c2w, c2w_staticcam=c2w_b, **kw)}
    for name, (rgb, disp, acc, extras) in cases.items():
        arrs[f"{name}/rgb"], arrs[f"{name}/disp"], arrs[f"{name}/acc"] = rgb, disp, acc
        for k in ("depth_map", "rgb0", "z_vals", "pred_hyp"):
            arrs[f"{name}/{k}"] = extras[k]
        arrs[f"{name}/n_extras"] = len(extras)
    for tag, p in (("coarse", pc), ("fine", pf)):
        for k, v in weight_digest(p).items():
            arrs[f"digest_{tag}/{k}"] = v
    save("f9_render_image", **arrs)


if __name__ == "__main__":
    f9_render_image()
    f8_rays()
    f1_embed()
    f2_mlp()
    f3_composite()
    f4_sample_pdf()
    f5_carve()
    f5_carve_knp()
    f7_perturb()
    f6_render()
    print("all fixtures written; oracle pinned bit-exact against the reference on this torch build")
