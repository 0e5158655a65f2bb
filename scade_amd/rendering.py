"""MI355X drop-in for the render operators that the reference keeps at module
level of its driver scripts (``run_scade_scannet.py`` / ``run_scade_wild.py``):

  batchify :39-46 · run_network :48-63 · batchify_rays :66-78 · render :80-155 ·
  render_hyp :157-233 · compute_weights :511-522 · raw2depth :524-528 ·
  raw2outputs :530-562 · perturb_z_vals :564-579 · render_rays :581-751

Same names / arguments / returned dict keys.  All arithmetic runs in
libscade_hip.so; torch is used for allocation, slicing and dict plumbing.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .ops import CoarseTailFn, CompositeFn, FineTailFn
from .run_nerf_helpers import (Embedder, NeRF, _draw_u, _sample, get_rays)


def batchify(fn, chunk):
    """run_scade_scannet.py:39-46."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


_BB_CACHE = {}


def _bb_tensor(bb_center, bb_scale, device):
    """{cx,cy,cz,scale} as one device tensor, cached per (center, scale) tensor OBJECT and version.
    The entry keeps both objects alive: an address (or id) can then never come back as a different
    tensor while its entry exists - a cache keyed on data_ptr alone returned a stale box once an
    earlier tensor's memory had been re-used."""
    if not torch.is_tensor(bb_center):
        bb_center = torch.as_tensor(bb_center, dtype=torch.float32)
    if not torch.is_tensor(bb_scale):
        bb_scale = torch.as_tensor(bb_scale, dtype=torch.float32)
    key = (id(bb_center), bb_center._version, id(bb_scale), bb_scale._version, str(device))
    hit = _BB_CACHE.get(key)
    if hit is None or hit[0] is not bb_center or hit[1] is not bb_scale:
        if len(_BB_CACHE) > 64:
            _BB_CACHE.clear()
        bb = torch.cat([bb_center.reshape(-1)[:3].float(), bb_scale.reshape(-1)[:1].float()]).to(device)
        hit = (bb_center, bb_scale, bb)
        _BB_CACHE[key] = hit
    return hit[2]


def _is_fusable(fn, embed_fn, embeddirs_fn, embedded_cam):
    net = fn.module if isinstance(fn, torch.nn.DataParallel) else fn
    if not isinstance(net, NeRF):
        return None
    if not (isinstance(embed_fn, Embedder) and embed_fn.multires == 9 and embed_fn.input_dims == 3):
        return None
    if not (isinstance(embeddirs_fn, Embedder) and embeddirs_fn.multires == 0):
        return None
    if embedded_cam is not None and embedded_cam.numel() != 0:
        return None
    return net


def run_network(inputs, viewdirs, embedded_cam, fn, embed_fn, embeddirs_fn, bb_center, bb_scale,
                netchunk=1024 * 64):
    """run_scade_scannet.py:48-63.  With the SCADE configuration (get_embedder(9),
    get_embedder(0), empty camera code, scade_amd.NeRF) the normalisation, both
    embeddings, the view broadcast and the 12 layers are ONE kernel launch and the
    [P,60] embedding never exists in HBM; ``netchunk`` (a memory knob of the
    reference) is then irrelevant."""
    net = _is_fusable(fn, embed_fn, embeddirs_fn, embedded_cam) if viewdirs is not None else None
    if net is not None and inputs.dim() == 3:
        return net.forward_points(inputs, viewdirs, _bb_tensor(bb_center, bb_scale, inputs.device))
    # generic composition (any callable fn / embedders)
    flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    flat = (flat - bb_center) * bb_scale
    embedded = embed_fn(flat)
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(inputs.shape)
        dirs = torch.reshape(dirs, [-1, dirs.shape[-1]])
        parts = [embedded, embeddirs_fn(dirs)]
        if embedded_cam is not None and embedded_cam.numel() != 0:
            parts.append(embedded_cam.unsqueeze(0).expand(dirs.shape[0], embedded_cam.shape[0]))
        embedded = torch.cat(parts, -1)
    out = batchify(fn, netchunk)(embedded)
    return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])


def make_network_query_fn(embed_fn, embeddirs_fn, bb_center, bb_scale, netchunk=1024 * 64):
    """The closure create_nerf builds (run_scade_scannet.py:461-466)."""
    def query(inputs, viewdirs, embedded_cam, network_fn):
        return run_network(inputs, viewdirs, embedded_cam, network_fn, embed_fn=embed_fn,
                           embeddirs_fn=embeddirs_fn, bb_center=bb_center, bb_scale=bb_scale,
                           netchunk=netchunk)
    return query


# ---------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------

def _noise_arg(noise, like):
    if torch.is_tensor(noise):
        return noise.to(like.device, torch.float32).expand(like.shape).contiguous()
    if noise == 0:
        return None
    return torch.full(like.shape, float(noise), device=like.device, dtype=torch.float32)


def compute_weights(raw, z_vals, rays_d, noise=0.):
    """run_scade_scannet.py:511-522."""
    return CompositeFn.apply(raw, z_vals, rays_d, _noise_arg(noise, raw[..., 3]))[3]


def raw2depth(raw, z_vals, rays_d):
    """run_scade_scannet.py:524-528 (tiny reductions on the kernel's weights)."""
    weights = compute_weights(raw, z_vals, rays_d)
    depth = torch.sum(weights * z_vals, -1)
    std = (((z_vals - depth.unsqueeze(-1)).pow(2) * weights).sum(-1)).sqrt()
    return depth, std


def _raw_noise(raw, raw_noise_std, pytest):
    """the density noise raw2outputs draws (:545-553), or None"""
    if not raw_noise_std > 0.:
        return None
    shape = raw[..., 3].shape
    if pytest:
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*list(shape)) * raw_noise_std).to(raw.device)
    return torch.randn(shape, device=raw.device) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, pytest=False):
    """run_scade_scannet.py:530-562 -> rgb_map, disp_map, acc_map, weights, depth_map."""
    return CompositeFn.apply(raw, z_vals, rays_d, _raw_noise(raw, raw_noise_std, pytest))


def perturb_z_vals(z_vals, pytest):
    """run_scade_scannet.py:564-579 on an arbitrary (already built) z_vals tensor.
    render_rays does not call this: its jitter is fused into scade_ray_points."""
    ops.check(z_vals, "perturb_z_vals: z_vals")
    if pytest:
        np.random.seed(0)
        t_rand = torch.Tensor(np.random.rand(*list(z_vals.shape))).to(z_vals.device)
    else:
        t_rand = torch.rand_like(z_vals)
    return ops.perturb_z(z_vals, t_rand)


# ---------------------------------------------------------------------------
# render_rays
# ---------------------------------------------------------------------------

def render_rays(ray_batch, use_viewdirs, network_fn, network_query_fn, N_samples,
                precomputed_z_samples=None, embedded_cam=None, retraw=False, lindisp=False,
                perturb=0., N_importance=0, network_fine=None, raw_noise_std=0., verbose=False,
                pytest=False, is_joint=False, cached_u=None, t_rand=None, u_coarse=None,
                coarse_stream=None, fuse_tails=True, draws=None, _stop_before_fine_tail=False, _coarse_pre=None):
    """run_scade_scannet.py:581-751 (live branch ``N_importance > 0``).

    Extra keyword-only knobs beyond the reference signature: ``t_rand`` [N,N_samples]
    and ``u_coarse`` [N,N_importance] inject the stratified jitter / first sampler
    draw (the reference can only inject the last draw through ``cached_u``); used by
    the parity tests because device RNG streams cannot match the CPU's.  ``coarse_stream`` (a
    ``torch.cuda.Stream``): run the coarse stage on that side stream.  The forward is unchanged
    (the fine stage waits for it), but autograd replays every backward node on its forward stream,
    so in a train step the whole coarse backward chain (composite -> dgrad -> wgrad) runs
    CONCURRENTLY with the fine chain instead of behind it and fills the tails of its launches.
    ``fuse_tails=False`` runs the per-ray work between the MLP launches as the separate
    raw2outputs / sample_pdf / merge operators (same bits; kept for the parity tests).
    ``draws`` (an ``ops.Draws``; training only, perturb > 0): the step's uniform draws that were not injected
    are made INSIDE the first kernel of the step (scade_ray_points_draw: counter-based Philox keyed by seed,
    step, ray and draw index) instead of by torch.rand launches.
    ``_stop_before_fine_tail`` (Trainer only): return after the fine MLP with ``raw`` / ``raw0`` / ``u`` and
    ``'_fine_tail_pending': True`` - the caller runs the fine tail, the loss and both tails' backward as ONE launch
    (ops.FineTailLossFn); ignored (a complete result is returned) when that launch cannot take the case.
    ``_coarse_pre`` (GraphedTrainer only; an ``ops.CoarsePoints``): the coarse samples - z_vals, positions, the samplers'
    draws - were already computed by the launch in front of the captured step (same kernel body, same bits)."""
    if N_importance <= 0:
        raise NotImplementedError(
            "render_rays: N_importance == 0 is dead code in the reference (raises UnboundLocalError "
            "'u' at run_scade_scannet.py:733); only the coarse+fine branch exists")
    if not use_viewdirs:
        raise NotImplementedError("render_rays: SCADE always renders with use_viewdirs=True")
    ops.check(ray_batch, "render_rays: ray_batch")
    ops.check_current_device(ray_batch, "render_rays: ray_batch")
    if ray_batch.dim() != 2 or ray_batch.shape[1] < 11:
        raise ValueError("render_rays: ray_batch must be [N, >=11] = o,d,near,far,viewdir")
    rays = ray_batch if ray_batch.stride(1) == 1 else ray_batch.contiguous()
    N = rays.shape[0]
    rays_d = rays[:, 3:6]
    viewdirs = rays[:, 8:11]
    dev = rays.device
    if embedded_cam is None:
        embedded_cam = torch.empty(0, device=dev)
    det = (perturb == 0.)

    # ---- coarse: z (+jitter) and points in one launch (:638-657) -------------
    in_kernel = perturb > 0. and t_rand is None and (draws is not None or _coarse_pre is not None) and not pytest
    if in_kernel:
        pass                                     # jitter drawn by scade_ray_points_draw (coarse_stage)
    elif perturb > 0.:
        if t_rand is None:
            if pytest:
                np.random.seed(0)
                t_rand = torch.Tensor(np.random.rand(N, N_samples)).to(dev)
            else:
                t_rand = torch.rand(N, N_samples, device=dev)
    else:
        t_rand = None
    # coarse stage = points -> MLP -> [raw2outputs -> detached importance samples (:702-711) -> sorted
    # merge + fine points (:713-714)]; the bracket is ONE launch (scade_ray_tail) when the row fits its
    # register sort, three otherwise
    fused = fuse_tails and ops.ray_tail_supported(N_samples, N_importance, merge=True)

    def coarse_stage():
        nonlocal u_coarse, cached_u
        if in_kernel and _coarse_pre is not None:
            z, p, ua, ub = _coarse_pre.z, _coarse_pre.pts, _coarse_pre.u_a, _coarse_pre.u_b
            u_coarse = ua if u_coarse is None else u_coarse
            cached_u = ub if (cached_u is None and not is_joint) else cached_u
        elif in_kernel:
            # the two samplers' draws that were not injected come out of the same launch (the last sampler's
            # stay with the host when is_joint shares ONE row among all rays, helpers:498-513)
            z, p, ua, ub = ops.ray_points_draw(rays, N_samples, lindisp, draws, N_importance,
                                               want_a=u_coarse is None, want_b=cached_u is None and not is_joint)
            u_coarse = ua if u_coarse is None else u_coarse
            cached_u = ub if cached_u is None else cached_u
        else:
            z, p = ops.ray_points(rays, N_samples, t_rand, lindisp)
        r = network_query_fn(p, viewdirs, embedded_cam, network_fn)
        if not fused:
            return (z, r) + tuple(raw2outputs(r, z, rays_d, raw_noise_std, pytest=pytest))
        noise = _raw_noise(r, raw_noise_std, pytest)
        # the coarse importance sampler is ALWAYS the per-ray sample_pdf (:705); only the last
        # sampler honours is_joint (:726-730)
        uc = u_coarse if u_coarse is not None else _draw_u(z, N_importance, det, pytest, False)
        return (z, r) + tuple(CoarseTailFn.apply(r, z, rays, noise, uc, N_importance))

    if coarse_stream is None:
        outs = coarse_stage()
    else:
        main = torch.cuda.current_stream()
        coarse_stream.wait_stream(main)
        with torch.cuda.stream(coarse_stream):
            outs = coarse_stage()
        main.wait_stream(coarse_stream)
        for t in outs:                      # allocated on the side stream, consumed on this one
            t.record_stream(main)
    z_vals_0, raw, rgb_map_0, disp_map_0, acc_map_0, weights_0, depth_map_0 = outs[:7]
    raw_0 = raw

    if fused:
        z_vals, pts = outs[7:]
    else:
        # ---- importance samples from the coarse pdf, detached (:702-711) ---------
        uc = u_coarse if u_coarse is not None else _draw_u(z_vals_0, N_importance, det, pytest, False)
        with torch.no_grad():
            z_samples = _sample(z_vals_0, weights_0[..., 1:-1], uc, bins_are_mids=True)
        # ---- merge + fine points (:713-714) --------------------------------------
        z_vals, pts = ops.merge_sorted(z_vals_0, z_samples, rays)
    run_fn = network_fn if network_fine is None else network_fine
    raw = network_query_fn(pts, viewdirs, embedded_cam, run_fn)

    # ---- fine stage: raw2outputs + depth hypotheses from the fine pdf (:720-730): one launch; when a
    # gradient is recorded also ONE backward launch (ops.FineTailFn)
    fusable = fuse_tails and ops.ray_tail_supported(z_vals.shape[1], N_importance, merge=False)
    if (_stop_before_fine_tail and fusable and torch.is_grad_enabled() and raw.requires_grad and raw_noise_std == 0.
            and z_vals.shape[1] <= 256 and z_vals_0.shape[1] <= 64):
        u = cached_u if cached_u is not None else _draw_u(z_vals, N_importance, det, pytest, is_joint)
        return {'_fine_tail_pending': True, 'raw': raw, 'raw0': raw_0, 'z_vals': z_vals, 'u': u, 'rgb0': rgb_map_0,
                'disp0': disp_map_0, 'acc0': acc_map_0, 'depth0': depth_map_0, 'z_vals0': z_vals_0,
                'weights0': weights_0}
    if fusable and not (torch.is_grad_enabled() and raw.requires_grad):
        noise = _raw_noise(raw, raw_noise_std, pytest)
        u = cached_u if cached_u is not None else _draw_u(z_vals, N_importance, det, pytest, is_joint)
        rgb_map, disp_map, acc_map, weights, depth_map, pred_depth_hyp, z_std, _, _ = ops.ray_tail(
            raw, z_vals, rays, noise, u, N_importance, merge=False, want_std=True)
    elif fusable:
        noise = _raw_noise(raw, raw_noise_std, pytest)
        u = cached_u if cached_u is not None else _draw_u(z_vals, N_importance, det, pytest, is_joint)
        rgb_map, disp_map, acc_map, weights, depth_map, pred_depth_hyp, z_std = FineTailFn.apply(
            raw, z_vals, rays, noise, u, N_importance)
    else:
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(
            raw, z_vals, rays_d, raw_noise_std, pytest=pytest)
        u = cached_u if cached_u is not None else _draw_u(z_vals, N_importance, det, pytest, is_joint)
        pred_depth_hyp, z_std = _sample(z_vals, weights[..., 1:-1], u, bins_are_mids=True, want_std=True)
    if u.dim() == 1 or u.stride(0) == 0:
        u = u.expand(N, N_importance)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map,
           'z_vals': z_vals, 'weights': weights, 'pred_hyp': pred_depth_hyp, 'u': u}
    if retraw:
        ret['raw'] = raw
    ret['rgb0'] = rgb_map_0
    ret['disp0'] = disp_map_0
    ret['acc0'] = acc_map_0
    ret['depth0'] = depth_map_0
    ret['z_vals0'] = z_vals_0
    ret['weights0'] = weights_0
    ret['z_std'] = z_std
    # (the reference's per-tensor isnan/isinf host syncs, :747-749, are DEBUG-only prints)
    return ret


_SIDE_STREAMS = {}


def _side_streams(device, n):
    key = (str(device), n)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE_STREAMS[key]


def batchify_rays(rays_flat, chunk=1024 * 32, use_viewdirs=False, streams=1, **kwargs):
    """run_scade_scannet.py:66-78.  ``streams`` > 1 (inference only) issues consecutive chunks
    on alternating HIP streams: ray chunks are independent, so the tail of one chunk's MLP launch
    overlaps the head of the next chunk's."""
    all_ret = {}
    if streams > 1 and not torch.is_grad_enabled() and rays_flat.shape[0] > chunk:
        cur = torch.cuda.current_stream(rays_flat.device)
        side = _side_streams(rays_flat.device, streams)
        # the lazily re-packed weight blobs are rebuilt by whichever launch first sees a changed
        # parameter: do that HERE, on the stream every side stream waits for, or chunk 1 could read a
        # blob that chunk 0's stream is still rewriting
        for net in (kwargs.get("network_fn"), kwargs.get("network_fine")):
            net = net.module if isinstance(net, torch.nn.DataParallel) else net
            if isinstance(net, NeRF):
                net.warm_packs()
        for st in side:
            st.wait_stream(cur)
        for n, i in enumerate(range(0, rays_flat.shape[0], chunk)):
            with torch.cuda.stream(side[n % streams]):
                ret = render_rays(rays_flat[i:i + chunk], use_viewdirs, **kwargs)
            for k in ret:
                ret[k].record_stream(cur)       # allocated on the side stream, consumed by cat() on cur
                all_ret.setdefault(k, []).append(ret[k])
        for st in side:
            cur.wait_stream(st)
    else:
        for i in range(0, rays_flat.shape[0], chunk):
            ret = render_rays(rays_flat[i:i + chunk], use_viewdirs, **kwargs)
            for k in ret:
                all_ret.setdefault(k, []).append(ret[k])
    return {k: torch.cat(all_ret[k], 0) for k in all_ret}


def _assemble_rays(H, W, intrinsic, rays, c2w, near, far, use_viewdirs, c2w_staticcam, rays_depth):
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, intrinsic, c2w)
    elif rays.shape[0] == 2:
        rays_o, rays_d = rays
    else:
        rays_o, rays_d, rays_depth = rays
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, intrinsic, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    sh = rays_d.shape
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    near, far = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
    parts = [rays_o, rays_d, near, far]
    if use_viewdirs:
        parts.append(viewdirs)
    if rays_depth is not None:
        parts.append(torch.reshape(rays_depth, [-1, 3]).float())
    return torch.cat(parts, -1), sh


def render(H, W, intrinsic, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1.,
           with_5_9=False, use_viewdirs=False, c2w_staticcam=None, rays_depth=None, streams=1,
           shard_group=None, shard_keys="image", **kwargs):
    """run_scade_scannet.py:80-155 (ray-row assembly is host plumbing).  ``with_5_9`` keeps the centre
    columns of a full-image render (:108-115, one third of 16:9, used by render_video).
    ``shard_group`` (a torch.distributed group, or True for the default group): the ray rows are split
    over the ranks and the per-pixel maps all-gathered (parallel.render_rays_sharded) - every rank returns
    the whole image; ``shard_keys`` = "image" (per-pixel maps only) or None (every key)."""
    if c2w is not None and use_viewdirs and c2w_staticcam is None and rays_depth is None:
        # full image: ray rows straight from the generation kernel (no [H,W,3] intermediates)
        rays_flat = ops.gen_rays(H, W, intrinsic, c2w, near=near, far=far)["rays"]
        sh = (H, W, 3)
    else:
        rays_flat, sh = _assemble_rays(H, W, intrinsic, rays, c2w, near, far, use_viewdirs,
                                       c2w_staticcam, rays_depth)
    if with_5_9 and c2w is not None:
        Wc = int(H / 9. * 16. / 3.)
        Wc -= Wc % 2
        start = (W - Wc) // 2
        rays_flat = rays_flat.reshape(H, W, -1)[:, start:start + Wc].reshape(H * Wc, -1).contiguous()
        sh = (H, Wc, 3)
    if shard_group is not None and shard_group is not False:
        from . import parallel
        all_ret = parallel.render_rays_sharded(
            rays_flat, lambda rows: batchify_rays(rows, chunk, use_viewdirs, streams=streams, **kwargs),
            group=None if shard_group is True else shard_group,
            keys=parallel.IMAGE_KEYS if shard_keys == "image" else shard_keys)
    else:
        all_ret = batchify_rays(rays_flat, chunk, use_viewdirs, streams=streams, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


# render_hyp (:157-233) is the same function body in the reference
render_hyp = render
