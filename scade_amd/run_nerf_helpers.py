"""MI355X drop-in for the hot-path operators of the reference's
``model/run_nerf_helpers.py``: same names, arguments and return values, every
numeric step executed by hand-written HIP kernels through ``libscade_hip.so``.

Mirrored interface (reference file:line):
  img2mse / mse2psnr / to8b / to16b            helpers:11-14
  compute_space_carving_loss                    helpers:93-128
  DenseLayer / Embedder / get_embedder / NeRF   helpers:131-247
  select_coordinates / get_ray_dirs / get_rays  helpers:279-305
  sample_pdf / sample_pdf_return_u /
  sample_pdf_joint / sample_pdf_joint_return_u  helpers:337-538

There is no CPU fallback: tensors must live on the HIP device.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import CarveFn, CarveJointShardedFn, MseFn, SamplePdfFn

# ---------------------------------------------------------------------------
# misc  (helpers:11-14)
# ---------------------------------------------------------------------------


def img2mse(x, y):
    return MseFn.apply(x, y, None)


def img2mse_masked(x, y, mask):
    """mean(((x-y)**2) * mask[:,None])  (run_scade_wild.py:978-986)."""
    return MseFn.apply(x, y, mask)


def mse2psnr(x):
    return -10.0 * torch.log(x) / torch.log(torch.full((1,), 10.0, device=x.device))


def to8b(x):
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def to16b(x):
    return ((2 ** 16 - 1) * np.clip(x, 0, 1)).astype(np.uint16)


# ---------------------------------------------------------------------------
# space-carving loss  (helpers:93-128)
# ---------------------------------------------------------------------------


def compute_space_carving_loss(pred_depth, target_hypothesis, is_joint=False, mask=None, norm_p=2,
                               threshold=0.0, sharded=False, group=None, n_total=None):
    """pred_depth [N,P]; target_hypothesis [K,N,1], or [K,N,P] when every sample (quantile) already
    picked its hypothesis (helpers:100-102).  The norm runs over a size-1 axis, so every ``norm_p``
    gives |pred - hyp| (kept for signature parity).

    ``sharded=True`` (not in the reference, which is single-process): the arguments are this
    rank's SHARD of a ray-partitioned batch.  Only ``is_joint=True`` needs an exchange (its mean
    over rays precedes the min over K): the result is then the loss of the whole batch, see
    ``ops.CarveJointShardedFn`` (``n_total`` = rays of all shards, None = found with one extra
    all-reduce)."""
    if target_hypothesis.dim() != 3:
        raise ValueError("compute_space_carving_loss: target_hypothesis must be [K,N,1] or [K,N,P]")
    if norm_p <= 0:
        raise ValueError("norm_p must be positive")
    if is_joint and sharded:
        return CarveJointShardedFn.apply(pred_depth, target_hypothesis, mask, float(threshold), group, n_total)
    return CarveFn.apply(pred_depth, target_hypothesis, mask, float(threshold), bool(is_joint))


# ---------------------------------------------------------------------------
# positional encoding  (helpers:142-189)
# ---------------------------------------------------------------------------


class Embedder:
    """gamma(x) = [x, sin(pi x 2^0), cos(pi x 2^0), ..., cos(pi x 2^(L-1))]."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not kwargs.get("include_input", True) or not kwargs.get("log_sampling", True):
            raise NotImplementedError("Embedder: only include_input=True, log_sampling=True "
                                      "(the get_embedder configuration) is implemented")
        self.multires = int(kwargs["num_freqs"])
        self.input_dims = int(kwargs["input_dims"])
        self.out_dim = self.input_dims * (1 + 2 * self.multires)

    def embed(self, inputs):
        return ops.embed(inputs, self.multires)

    __call__ = embed


def get_embedder(multires, i=0):
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return eo, eo.out_dim


# ---------------------------------------------------------------------------
# NeRF MLP  (helpers:131-139, 193-247)
# ---------------------------------------------------------------------------


class DenseLayer(nn.Linear):
    def __init__(self, in_dim, out_dim, activation="relu", *args, **kwargs):
        self.activation = activation
        super().__init__(in_dim, out_dim, *args, **kwargs)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain(self.activation))
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class NeRF(nn.Module):
    """Parameter container + fused HIP forward.  state_dict names match the
    reference (helpers:205-220); checkpoints saved through nn.DataParallel
    ('module.' prefix, run_scade_scannet.py:438) load via ``load_reference_state_dict``."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, input_ch_cam=0, output_ch=4,
                 skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views, self.input_ch_cam = input_ch, input_ch_views, input_ch_cam
        self.skips, self.use_viewdirs = list(skips), use_viewdirs
        self.pts_linears = nn.ModuleList(
            [DenseLayer(input_ch, W)] +
            [DenseLayer(W + input_ch if i in self.skips else W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([DenseLayer(input_ch_views + input_ch_cam + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = DenseLayer(W, W, activation="linear")
            self.alpha_linear = DenseLayer(W, 1, activation="linear")
            self.rgb_linear = DenseLayer(W // 2, 3, activation="linear")
        else:
            self.output_linear = DenseLayer(W, output_ch, activation="linear")
        self._packed = None
        self._packed_key = None
        # "f32": exact fp32 MFMA everywhere.  "f16x3": no-grad forwards use the split-precision
        # kernel (two fp16 planes per value, three f16 MFMAs per product; ~1e-6 relative error).
        # "f16" / "bf16": single-plane 16-bit operands, fp32 accumulate (ordinary mixed precision,
        # ~1e-3 / ~1e-2 relative error: BASELINE.json config 5's "bf16 MFMA path")
        self.inference_precision = "f32"
        # "f32": exact training kernels.  "f16x3": forward, dgrad and wgrad on the split-precision
        # kernels ("f16x3-dgrad": wgrad stays exact fp32).  "bf16-s8": "bf16" with the rows saved for the weight
        # gradient (activations, dZ) kept as 8-bit e5m2 in HBM (half the bytes of the HBM-bound training step;
        # forward and dgrad arithmetic unchanged).  "f16" / "bf16": mixed-precision training,
        # 16-bit activations / gradients / weight copies, fp32 accumulate and fp32 master weights
        self.train_precision = "f32"

    # -- kernel support ----------------------------------------------------
    def _require_supported(self):
        ok = (self.D == 8 and self.W == 256 and self.input_ch == 57 and self.input_ch_views == 3
              and self.input_ch_cam == 0 and self.skips == [4] and self.use_viewdirs)
        if not ok:
            raise NotImplementedError(
                "scade_amd.NeRF: the HIP kernels implement the SCADE configuration "
                "NeRF(D=8, W=256, input_ch=57, input_ch_views=3, input_ch_cam=0, skips=[4], "
                "use_viewdirs=True) only")

    def ordered_params(self):
        """The 24 parameters in kernel order.  Cached: the walk over named_parameters() costs ~60 us
        and this is called a dozen times per step; the cache is keyed on the identity of the first
        and last layer modules and of their Parameter objects, which change if a layer or a parameter
        is ever re-assigned.  (Validated through the modules' own dicts: nn.Module.__getattr__ and
        ModuleList.__getitem__ are Python-level and cost 4.5 us per call, 60 us per eager step.)"""
        c = self.__dict__.get("_ordered_cache")
        if c is not None:
            mods = self._modules
            first_m = mods["pts_linears"]._modules["0"]
            last_m = mods["rgb_linear" if self.use_viewdirs else "output_linear"]
            if first_m is c[3] and last_m is c[4] and first_m._parameters["weight"] is c[0] \
                    and last_m._parameters["bias"] is c[1]:
                return c[2]
        first_m = self.pts_linears[0]
        last_m = self.rgb_linear if self.use_viewdirs else self.output_linear
        sd = dict(self.named_parameters())
        c = (first_m.weight, last_m.bias, [sd[k] for k in ops.PARAM_ORDER], first_m, last_m)
        self.__dict__["_ordered_cache"] = c
        return c[2]

    def packed(self):
        """MFMA-ordered parameter blob, rebuilt whenever a parameter changed in place."""
        self._require_supported()
        ps = self.ordered_params()
        key = (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key or self._packed.device != ps[0].device:
            self._packed = ops.mlp_pack(ps, None if self._packed is None or
                                        self._packed.device != ps[0].device else self._packed)
            self._packed_key = key
        return self._packed

    def packed_t(self):
        """Transposed weight pack for the backward, same invalidation rule as packed()."""
        ps = self.ordered_params()
        key = (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_t", None) is None or key != self._packed_t_key \
                or self._packed_t.device != ps[0].device:
            self._packed_t = ops.mlp_pack_t(ps)
            self._packed_t_key = key
        return self._packed_t

    def packed_f16(self):
        ps = self.ordered_params()
        key = (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_f16", None) is None or key != self._packed_f16_key \
                or self._packed_f16.device != ps[0].device:
            self._packed_f16 = ops.mlp_pack_f16(ps)
            self._packed_f16_key = key
        return self._packed_f16

    def packed_t_f16(self):
        ps = self.ordered_params()
        key = (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_t_f16", None) is None or key != self._packed_t_f16_key \
                or self._packed_t_f16.device != ps[0].device:
            self._packed_t_f16 = ops.mlp_pack_t_f16(ps)
            self._packed_t_f16_key = key
        return self._packed_t_f16

    def packed_lp(self, bf16):
        """Single-plane fp16 / bf16 weight pack of the 16-bit inference kernel (same invalidation rule)."""
        ps = self.ordered_params()
        key = (bool(bf16), ops.PARAM_EPOCH) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_lp", None) is None or key != self._packed_lp_key \
                or self._packed_lp.device != ps[0].device:
            self._packed_lp = ops.mlp_pack_lp(ps, bf16)
            self._packed_lp_key = key
        return self._packed_lp

    def packed_t_lp(self, bf16):
        ps = self.ordered_params()
        key = (bool(bf16), ops.PARAM_EPOCH) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_packed_t_lp", None) is None or key != self._packed_t_lp_key \
                or self._packed_t_lp.device != ps[0].device:
            self._packed_t_lp = ops.mlp_pack_t_lp(ps, bf16)
            self._packed_t_lp_key = key
        return self._packed_t_lp

    # -- the Trainer's one-launch pack of both networks (ops.mlp_pack_step) ------------------------------
    def pack_key(self):
        return (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version) for p in self.ordered_params())

    def pack_stale(self, fmt, transposed, key=None):
        """Is the cached training pack of this format stale? (fmt: "f32" | "bf16" | "f16")"""
        key = self.pack_key() if key is None else key
        dev = self.ordered_params()[0].device
        if fmt == "f32":
            blob, k = (getattr(self, "_packed_t", None), getattr(self, "_packed_t_key", None)) if transposed \
                else (self._packed, self._packed_key)
            return blob is None or k != key or blob.device != dev
        blob, k = (getattr(self, "_packed_t_lp", None), getattr(self, "_packed_t_lp_key", None)) if transposed \
            else (getattr(self, "_packed_lp", None), getattr(self, "_packed_lp_key", None))
        return blob is None or k != (fmt == "bf16",) + key or blob.device != dev

    def adopt_packs(self, fmt, fwd, transposed, key):
        """Take over blobs ops.mlp_pack_step wrote for the parameter state ``key`` describes.  (Plain attributes,
        written through the instance dict: nn.Module.__setattr__ costs 3 us per assignment.)"""
        d = self.__dict__
        if fmt == "f32":
            if fwd is not None:
                d["_packed"], d["_packed_key"] = fwd, key
            if transposed is not None:
                d["_packed_t"], d["_packed_t_key"] = transposed, key
        else:
            k = (fmt == "bf16",) + key
            if fwd is not None:
                d["_packed_lp"], d["_packed_lp_key"] = fwd, k
            if transposed is not None:
                d["_packed_t_lp"], d["_packed_t_lp_key"] = transposed, k

    def warm_packs(self):
        """Bring the weight pack of the current inference precision up to date on the CURRENT stream
        (callers that fan work out over several streams do this before forking)."""
        if self.inference_precision == "f16x3":
            self.packed_f16()
        elif self.inference_precision in ("f16", "bf16"):
            self.packed_lp(self.inference_precision == "bf16")
        else:
            self.packed()

    INFERENCE_PRECISIONS = ("f32", "f16x3", "f16", "bf16")
    TRAIN_PRECISIONS = ("f32", "f16x3", "f16x3-dgrad", "f16", "bf16", "bf16-s8")

    def _fast(self, train):
        if self.inference_precision not in self.INFERENCE_PRECISIONS:
            raise ValueError("NeRF.inference_precision must be one of %s" % (self.INFERENCE_PRECISIONS,))
        return self.inference_precision != "f32" and not train

    def _fast_forward(self, inp, viewdirs, bb):
        if self.inference_precision == "f16x3":
            return ops.mlp_fwd_f16(self.packed_f16(), inp, viewdirs, bb)
        bf16 = self.inference_precision == "bf16"
        return ops.mlp_fwd_lp(self.packed_lp(bf16), bf16, inp, viewdirs, bb)

    def forward(self, x):
        """x [P, 60] = [gamma(pts) | viewdir] -> [P,4] (helpers:223-247)."""
        from .mlp import MlpEmbeddedFn
        self._require_supported()
        ops.check_current_device(x, "NeRF.forward: x")
        ps = self.ordered_params()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in ps)
        if self._fast(train):
            return self._fast_forward(x, None, None)
        return MlpEmbeddedFn.apply(self, train, x, *ps)

    def forward_points(self, pts, viewdirs, bb):
        """Fused run_network: pts [N,S,3], viewdirs [N,3], bb [4]={center,scale} -> raw [N,S,4]."""
        from .mlp import MlpPointsFn
        self._require_supported()
        ps = self.ordered_params()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in ps)
        if self._fast(train):
            return self._fast_forward(pts, viewdirs, bb)
        if train and self.__dict__.get("_grad_sink") is not None:
            # FlatParams.attach_grad_sinks: the backward writes the flat gradient itself and hands autograd nothing,
            # so ONE parameter is enough to put the node into the graph (29 arguments through Function.apply and
            # 24 gradient edges per network were 40 us of an eager step)
            return MlpPointsFn.apply(self, train, pts, viewdirs, bb, next(p for p in ps if p.requires_grad))
        return MlpPointsFn.apply(self, train, pts, viewdirs, bb, *ps)

    def load_reference_state_dict(self, state_dict, strict=True):
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return self.load_state_dict(sd, strict=strict)


# ---------------------------------------------------------------------------
# rays  (helpers:279-305) -- host-side batch assembly, not on the kernel path
# ---------------------------------------------------------------------------


def select_coordinates(coords, N_rand):
    coords = torch.reshape(coords, [-1, 2])
    select_inds = np.random.choice(coords.shape[0], size=[N_rand], replace=False)
    return coords[select_inds].long()


def _coords_rc(coords):
    """reference coords are [N,2] = (row, col), float (meshgrid of linspace) or long"""
    return None if coords is None else coords.reshape(-1, 2).to(torch.int32)


def get_ray_dirs(H, W, intrinsic, c2w, coords=None):
    """helpers:285-299 (scade_gen_rays)."""
    out = ops.gen_rays(H, W, intrinsic, c2w, coords=_coords_rc(coords), want_rows=False, want_od=True)
    return out["rays_d"] if coords is not None else out["rays_d"].reshape(H, W, 3)


def get_rays(H, W, intrinsic, c2w, coords=None):
    """helpers:301-305 -> (rays_o, rays_d); (H,W,3) each without coords, (N,3) with."""
    out = ops.gen_rays(H, W, intrinsic, c2w, coords=_coords_rc(coords), want_rows=False, want_od=True)
    if coords is not None:
        return out["rays_o"], out["rays_d"]
    return out["rays_o"].reshape(H, W, 3), out["rays_d"].reshape(H, W, 3)


def get_ray_batch(H, W, intrinsic, c2w, coords, near, far, image=None, hypotheses=None,
                  mask_corners=False, mask_edges=False):
    """The gathers of get_ray_batch_from_one_image_hypothesis_idx (run_scade_scannet.py:784-821)
    fused with the ray-row assembly of render_hyp (:200-219): only the N selected pixels are
    touched (the reference generates all H*W rays every step).
      image [H,W,3]; hypotheses [K,H,W] or [K,H,W,1]
    -> rays [N,11], target_s [N,3] | None, target_h [K,N,1] | None, mask [N] | None"""
    hy = None if hypotheses is None else hypotheses.reshape(hypotheses.shape[0], H, W)
    # (the corner mask wins when both are asked for: the edge mask is an ``elif``, run_scade_wild.py:805,818)
    out = ops.gen_rays(H, W, intrinsic, c2w, coords=_coords_rc(coords), near=near, far=far, image=image,
                       hyps=hy, corner_px=20 if mask_corners else 0, edge_px=10 if (mask_edges and not mask_corners) else 0,
                       want_rows=True, want_mask=bool(mask_corners or mask_edges))
    th = None if out["target_h"] is None else out["target_h"].unsqueeze(-1)
    return out["rays"], out["target_s"], th, out["mask"]


# ---------------------------------------------------------------------------
# hierarchical sampling  (helpers:337-538)
# ---------------------------------------------------------------------------


def _draw_u(bins, N_samples, det, pytest, joint):
    """The u the reference draws (helpers:346-361, 449-464); device RNG when random."""
    n_rays = bins.shape[0]
    dev = bins.device
    if pytest:
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0.0, 1.0, N_samples), [n_rays, N_samples])
        else:
            u = np.random.rand(n_rays, N_samples)
        return torch.Tensor(np.ascontiguousarray(u)).to(dev)
    if det:
        return ops.linspace01(N_samples, dev).expand(n_rays, N_samples)
    if joint:
        return torch.rand(N_samples, device=dev).expand(n_rays, N_samples)
    return torch.rand(n_rays, N_samples, device=dev)


def _sample(bins, weights, u, bins_are_mids=False, want_std=False):
    return SamplePdfFn.apply(bins, weights, u, bins_are_mids, want_std)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    return _sample(bins, weights, _draw_u(bins, N_samples, det, pytest, False))


def sample_pdf_return_u(bins, weights, N_samples, det=False, pytest=False, load_u=None):
    u = _draw_u(bins, N_samples, det, pytest, False) if load_u is None else load_u
    return _sample(bins, weights, u), u


def sample_pdf_joint(bins, weights, N_samples, det=False, pytest=False):
    return _sample(bins, weights, _draw_u(bins, N_samples, det, pytest, True))


def sample_pdf_joint_return_u(bins, weights, N_samples, det=False, pytest=False, load_u=None):
    u = _draw_u(bins, N_samples, det, pytest, True) if load_u is None else load_u
    return _sample(bins, weights, u), u
