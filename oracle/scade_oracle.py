"""CPU oracle for the SCADE per-ray rendering hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``scade_amd``) never routes through this module and
fails loudly when the HIP library is missing.

It is a PyTorch-CPU restatement of the reference algorithm (the reference is
pure Python/PyTorch, so the restatement is too; it issues the same ATen ops in
the same order so that, on the same torch build, it is bit-exact against the
imported reference).  Parity pinning: ``tools/make_golden.py`` imports the
reference from ``/root/reference`` (survey container only), asserts this
module equals it on every fixture, and writes the fixtures under
``tests/golden/``.  The reference ships no tests / golden vectors of its own
(SURVEY.md section 4), so those fixtures are the pin; ``tools/fuzz_oracle_vs_reference.py``
repeats the bit-for-bit comparison at random shapes and flags (15,400 comparisons,
``profiles/r06_oracle_vs_reference.json``).  (The checkers under ``tools/`` - the
golden generator, the randomized differential runs the GPU tests take slices of -
use this module the way the tests do: as the checker, never as the thing measured.)

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# losses  (model/run_nerf_helpers.py:11-12, 93-128)
# ----------------------------------------------------------------------------


def img2mse(x: Tensor, y: Tensor) -> Tensor:
    """model/run_nerf_helpers.py:11"""
    return torch.mean((x - y) ** 2)


def mse2psnr(x: Tensor) -> Tensor:
    """model/run_nerf_helpers.py:12"""
    return -10.0 * torch.log(x) / torch.log(torch.full((1,), 10.0))


def compute_space_carving_loss(pred_depth: Tensor, target_hypothesis: Tensor,
                               is_joint: bool = False, mask: Optional[Tensor] = None,
                               norm_p: int = 2, threshold: float = 0.0) -> Tensor:
    """model/run_nerf_helpers.py:93-128.

    pred_depth [N,P]; target_hypothesis [K,N,1] (or [K,N,P] when quantiles were
    cached).  The norm is over a trailing size-1 axis, i.e. |pred - hyp|.
    """
    n_pts = pred_depth.shape[1]
    hyp = target_hypothesis
    if hyp.shape[-1] == 1:
        hyp = hyp.repeat(1, 1, n_pts)                                        # :99
    dist = torch.norm(pred_depth.unsqueeze(-1) - hyp.unsqueeze(-1), p=norm_p, dim=-1)  # :106
    if mask is not None:                                                     # :108-110
        dist = dist * mask.unsqueeze(0).repeat(dist.shape[0], 1).unsqueeze(-1)
    if threshold > 0:                                                        # :112-113
        dist = torch.where(dist < threshold, torch.tensor([0.0]), dist)
    if is_joint:                                                             # :115-119
        per_hyp = torch.mean(dist, dim=1)
        best = torch.min(per_hyp, dim=0)[0]
        return torch.mean(best, dim=-1)
    best = torch.min(dist, dim=0)[0]                                         # :124
    return torch.mean(torch.mean(best, dim=-1))                              # :125-126


# ----------------------------------------------------------------------------
# positional encoding  (model/run_nerf_helpers.py:141-189)
# ----------------------------------------------------------------------------


def embed(x: Tensor, multires: int) -> Tensor:
    """gamma(x) = [x, sin(x*pi*2^0), cos(x*pi*2^0), ..., sin/cos(x*pi*2^(L-1))].

    Follows Embedder.create_embedding_fn / embed (helpers:146-172): the argument
    is ``(x * np.pi) * freq`` with freq a 0-dim fp32 tensor from
    ``2.**linspace(0, L-1, L)``.  ``multires == 0`` is the identity (3 ch).
    """
    parts = [x]
    if multires > 0:
        freqs = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires)
        for f in freqs:
            parts.append(torch.sin(x * np.pi * f))
            parts.append(torch.cos(x * np.pi * f))
    return torch.cat(parts, -1)


def embed_dim(multires: int) -> int:
    return 3 + 6 * multires


# ----------------------------------------------------------------------------
# NeRF MLP  (model/run_nerf_helpers.py:131-139, 193-247)
# ----------------------------------------------------------------------------

D_LAYERS, WIDTH, SKIP = 8, 256, 4


def nerf_param_shapes(input_ch: int = 57, input_ch_views: int = 3, W: int = WIDTH,
                      D: int = D_LAYERS) -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of NeRF(D=8,W=256,skips=[4],use_viewdirs=True)
    (helpers:205-220); [out,in] row-major like nn.Linear."""
    shp: Dict[str, Tuple[int, ...]] = {}
    for i in range(D):
        k = input_ch if i == 0 else (W + input_ch if (i - 1) == SKIP else W)
        shp[f"pts_linears.{i}.weight"] = (W, k)
        shp[f"pts_linears.{i}.bias"] = (W,)
    shp["views_linears.0.weight"] = (W // 2, input_ch_views + W)
    shp["views_linears.0.bias"] = (W // 2,)
    shp["feature_linear.weight"] = (W, W)
    shp["feature_linear.bias"] = (W,)
    shp["alpha_linear.weight"] = (1, W)
    shp["alpha_linear.bias"] = (1,)
    shp["rgb_linear.weight"] = (3, W // 2)
    shp["rgb_linear.bias"] = (3,)
    return shp


_RELU_LAYERS = ("pts_linears", "views_linears")


def nerf_init(seed: int, input_ch: int = 57, input_ch_views: int = 3) -> Dict[str, Tensor]:
    """Xavier-uniform init, gain sqrt(2) for ReLU layers / 1 for linear heads,
    zero bias (DenseLayer.reset_parameters, helpers:136-139).  The draw order is
    this module's own (not the reference's nn.Module construction order)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in nerf_param_shapes(input_ch, input_ch_views).items():
        if name.endswith(".bias"):
            out[name] = torch.zeros(shape)
        else:
            gain = math.sqrt(2.0) if name.startswith(_RELU_LAYERS) else 1.0
            bound = gain * math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
    return out


def nerf_forward(p: Dict[str, Tensor], x: Tensor, input_ch: int = 57) -> Tensor:
    """NeRF.forward with use_viewdirs=True (helpers:223-247).  x [P, input_ch+3]
    -> [P,4] = [rgb (pre-sigmoid), softplus(alpha, beta=10)]."""
    pts, views = x[..., :input_ch], x[..., input_ch:]
    h = pts
    for i in range(D_LAYERS):
        h = F.relu(F.linear(h, p[f"pts_linears.{i}.weight"], p[f"pts_linears.{i}.bias"]))
        if i == SKIP:
            h = torch.cat([pts, h], -1)                                      # :229-230
    alpha = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])   # :233
    feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])  # :234
    h = torch.cat([feat, views], -1)                                         # :235
    h = F.relu(F.linear(h, p["views_linears.0.weight"], p["views_linears.0.bias"]))
    rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])         # :241
    return torch.cat([rgb, F.softplus(alpha, beta=10)], -1)                  # :242


def run_network(pts: Tensor, viewdirs: Tensor, fn: Callable[[Tensor], Tensor],
                bb_center: Tensor, bb_scale: Tensor, multires: int = 9,
                netchunk: int = 1024 * 64) -> Tensor:
    """run_scade_scannet.py:48-63 (with embedded_cam empty, multires_views=0)."""
    flat = torch.reshape(pts, [-1, pts.shape[-1]])
    flat = (flat - bb_center) * bb_scale                                     # :52
    emb = embed(flat, multires)                                              # :53
    dirs = viewdirs[:, None].expand(pts.shape)                               # :56
    dirs = torch.reshape(dirs, [-1, dirs.shape[-1]])
    emb = torch.cat([emb, dirs], -1)                                         # :59 (cam width 0)
    out = torch.cat([fn(emb[i:i + netchunk]) for i in range(0, emb.shape[0], netchunk)], 0)
    return torch.reshape(out, list(pts.shape[:-1]) + [out.shape[-1]])


# ----------------------------------------------------------------------------
# alpha compositing  (run_scade_scannet.py:511-562)
# ----------------------------------------------------------------------------


def compute_weights(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise=0.0) -> Tensor:
    """run_scade_scannet.py:511-522."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 3] + noise) * dists)
    trans = torch.cumprod(
        torch.cat([torch.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    return alpha * trans


def raw2outputs(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise=0.0):
    """run_scade_scannet.py:530-562 -> rgb_map, disp_map, acc_map, weights, depth_map.
    ``noise`` is the already-drawn sigma noise tensor (or 0)."""
    rgb = torch.sigmoid(raw[..., :3])
    weights = compute_weights(raw, z_vals, rays_d, noise)
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map),
                               depth_map / torch.sum(weights, -1))
    acc_map = torch.sum(weights, -1)
    return rgb_map, disp_map, acc_map, weights, depth_map


def perturb_z_vals(z_vals: Tensor, t_rand: Tensor) -> Tensor:
    """run_scade_scannet.py:564-579 with the uniform draw passed in."""
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


# ----------------------------------------------------------------------------
# inverse-CDF sampler  (model/run_nerf_helpers.py:337-538)
# ----------------------------------------------------------------------------


def pdf_to_cdf(weights: Tensor) -> Tensor:
    """helpers:339-343: cdf = [0, cumsum((w+1e-5)/sum(w+1e-5))]."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)


def invert_cdf(bins: Tensor, cdf: Tensor, u: Tensor):
    """helpers:363-381.  Returns (samples, inds) with inds the int64
    searchsorted(right=True) result."""
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0]), inds


def draw_u(n_rays: int, n_samples: int, det: bool, pytest: bool = False,
           joint: bool = False) -> Tensor:
    """The u the reference would draw (helpers:346-361 / 449-464): linspace
    when det; numpy-seeded when pytest; else torch.rand ([S] shared by all rays
    when joint, expanded by broadcasting in searchsorted)."""
    if pytest:
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0.0, 1.0, n_samples), [n_rays, n_samples])
        else:
            u = np.random.rand(n_rays, n_samples)
        return torch.Tensor(u)
    if det:
        return torch.linspace(0.0, 1.0, steps=n_samples).expand([n_rays, n_samples])
    if joint:
        return torch.rand(n_samples).expand([n_rays, n_samples])
    return torch.rand([n_rays, n_samples])


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """helpers:337-383 with u explicit."""
    return invert_cdf(bins, pdf_to_cdf(weights), u)[0]


# ----------------------------------------------------------------------------
# render_rays  (run_scade_scannet.py:581-751, live branch N_importance > 0)
# ----------------------------------------------------------------------------


def render_rays(ray_batch: Tensor, coarse: Dict[str, Tensor], fine: Dict[str, Tensor],
                bb_center: Tensor, bb_scale: Tensor, n_samples: int = 64,
                n_importance: int = 128, t_rand: Optional[Tensor] = None,
                u_coarse: Optional[Tensor] = None, u_fine: Optional[Tensor] = None,
                lindisp: bool = False, retraw: bool = False,
                netchunk: int = 1024 * 64) -> Dict[str, Tensor]:
    """ray_batch [N,11] = o(3) d(3) near far viewdir(3).

    ``t_rand`` [N,n_samples] is the stratified jitter (None == perturb 0);
    ``u_coarse`` / ``u_fine`` [N,n_importance] are the sampler draws (None ==
    det linspace).  Returns the reference's dict (:733-744)."""
    N = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, 8:11]
    bounds = torch.reshape(ray_batch[..., 6:8], [-1, 1, 2])
    near, far = bounds[..., 0], bounds[..., 1]
    t_vals = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z_vals = near * (1.0 - t_vals) + far * t_vals                         # :642
    else:
        z_vals = 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)     # :645
    if t_rand is not None:
        z_vals = perturb_z_vals(z_vals, t_rand)                               # :655

    def query(pts, params):
        return run_network(pts, viewdirs, lambda e: nerf_forward(params, e),
                           bb_center, bb_scale, netchunk=netchunk)

    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]  # :657
    raw0 = query(pts, coarse)
    rgb0, disp0, acc0, w0, depth0 = raw2outputs(raw0, z_vals, rays_d)         # :660
    z0 = z_vals

    z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])                        # :702
    uc = draw_u(N, n_importance, det=True) if u_coarse is None else u_coarse
    z_samples = sample_pdf(z_mid, w0[..., 1:-1], uc).detach()                 # :705-711
    z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)            # :713
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]  # :714
    raw = query(pts, fine)                                                    # :718
    rgb, disp, acc, w, depth = raw2outputs(raw, z_vals, rays_d)               # :720

    z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])                        # :723
    uf = draw_u(N, n_importance, det=True) if u_fine is None else u_fine
    pred_hyp = sample_pdf(z_mid, w[..., 1:-1], uf)                            # :726

    ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth,
           "z_vals": z_vals, "weights": w, "pred_hyp": pred_hyp, "u": uf,
           "rgb0": rgb0, "disp0": disp0, "acc0": acc0, "depth0": depth0,
           "z_vals0": z0, "weights0": w0,
           "z_std": torch.std(pred_hyp, dim=-1, unbiased=False)}              # :744
    if retraw:
        ret["raw"] = raw
    return ret


def train_loss(ret: Dict[str, Tensor], target_s: Tensor, target_h: Tensor,
               space_carving_weight: float = 0.007, mask: Optional[Tensor] = None,
               norm_p: int = 2, threshold: float = 0.0, is_joint: bool = False):
    """run_scade_scannet.py:968-983: mse(rgb) + w*carve + mse(rgb0).  Returns
    (loss, img_loss, carve, img_loss0)."""
    img_loss = img2mse(ret["rgb_map"], target_s)
    carve = compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=is_joint,
                                       mask=mask, norm_p=norm_p, threshold=threshold)
    img_loss0 = img2mse(ret["rgb0"], target_s)
    loss = img_loss + space_carving_weight * carve + img_loss0
    return loss, img_loss, carve, img_loss0


# ----------------------------------------------------------------------------
# ray generation  (model/run_nerf_helpers.py:285-305; run_scade_scannet.py:122-141)
# ----------------------------------------------------------------------------


def get_rays(H: int, W: int, intrinsic: Tensor, c2w: Tensor, coords: Optional[Tensor] = None):
    """helpers:285-305: pixel-centre rays, OpenGL -z convention.  coords [N,2] = (row, col)."""
    fx, fy, cx, cy = intrinsic[0], intrinsic[1], intrinsic[2], intrinsic[3]
    if coords is None:
        i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
        i, j = i.t(), j.t()
    else:
        i, j = coords[:, 1], coords[:, 0]
    dirs = torch.stack([((i + 0.5) - cx) / fx, (H - (j + 0.5) - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ray_rows(rays_o: Tensor, rays_d: Tensor, near: float, far: float) -> Tensor:
    """render()/render_hyp() row assembly with use_viewdirs (run_scade_scannet.py:122-141)."""
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    nf = torch.ones_like(rays_d[..., :1])
    return torch.cat([rays_o, rays_d, near * nf, far * nf, viewdirs], -1)


# ----------------------------------------------------------------------------
# synthetic workload shared by tests / bench
# ----------------------------------------------------------------------------


def synthetic_rays(n_rays: int, seed: int = 0, near: float = 0.1, far: float = 5.0,
                   unit_dirs: bool = True) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    o = 0.1 * torch.randn(n_rays, 3, generator=g)
    d = torch.randn(n_rays, 3, generator=g)
    vd = d / torch.norm(d, dim=-1, keepdim=True)
    if unit_dirs:
        d = vd
    nf = torch.tensor([near, far]).expand(n_rays, 2)
    return torch.cat([o, d, nf, vd], -1).contiguous()
