"""autograd wrappers of the fused NeRF MLP kernels (scade_mlp_fwd / scade_mlp_bwd)."""
from __future__ import annotations

import torch

from . import ops



def _check_train_precision(net):
    if net.train_precision not in net.TRAIN_PRECISIONS:
        raise ValueError("NeRF.train_precision must be one of %s" % (net.TRAIN_PRECISIONS,))


def _needs_grad(params) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)


class MlpEmbeddedFn(torch.autograd.Function):
    """NeRF.forward(x[P,60])  (model/run_nerf_helpers.py:223-247)."""

    @staticmethod
    def forward(ctx, net, train, x, *params):
        acts = None
        lp = train and net.train_precision in ops.LP_FORMATS
        if train:
            _check_train_precision(net)
            acts = (ops.mlp_acts_lp_alloc if lp else ops.mlp_acts_alloc)(x.shape[0], x.device)
        if lp:
            code, bf16 = ops.LP_FORMATS[net.train_precision]
            out = ops.mlp_fwd_lp(net.packed_lp(bf16), code, x, None, None, acts)
        elif train and net.train_precision in ("f16x3", "f16x3-dgrad"):
            out = ops.mlp_fwd_f16(net.packed_f16(), x, None, None, acts, rows24=net.train_precision == "f16x3")
        else:
            out = ops.mlp_fwd_embedded(net.packed(), x, acts)
        ctx.net, ctx.mode = net, 0
        ctx.save_for_backward(x, acts if acts is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, g_out):
        from .mlp_bwd import mlp_backward
        x, acts = ctx.saved_tensors
        grads = mlp_backward(ctx.net, acts, g_out)
        return (None, None, None) + tuple(grads)


class MlpPointsFn(torch.autograd.Function):
    """run_network fused with the positional encoding (run_scade_scannet.py:48-63)."""

    @staticmethod
    def forward(ctx, net, train, pts, viewdirs, bb, *params):
        acts = None
        lp = train and net.train_precision in ops.LP_FORMATS
        if train:
            _check_train_precision(net)
            P = pts.shape[0] * pts.shape[1]
            acts = (ops.mlp_acts_lp_alloc if lp else ops.mlp_acts_alloc)(P, pts.device)
        if lp:
            code, bf16 = ops.LP_FORMATS[net.train_precision]
            out = ops.mlp_fwd_lp(net.packed_lp(bf16), code, pts, viewdirs, bb, acts)
        elif train and net.train_precision in ("f16x3", "f16x3-dgrad"):
            out = ops.mlp_fwd_f16(net.packed_f16(), pts, viewdirs, bb, acts, rows24=net.train_precision == "f16x3")
        else:
            out = ops.mlp_fwd_points(net.packed(), pts, viewdirs, bb, acts)
        ctx.net, ctx.mode, ctx.n_params = net, 1, len(params)
        ctx.save_for_backward(pts, viewdirs, bb, acts if acts is not None else pts.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, g_out):
        from .mlp_bwd import mlp_backward
        pts, viewdirs, bb, acts = ctx.saved_tensors
        grads = mlp_backward(ctx.net, acts, g_out)
        if ctx.n_params != len(grads):
            # the forward saw a gradient sink and passed one stand-in parameter (NeRF.forward_points): the sink has
            # taken the gradient and there is nothing to return - unless the sink was detached in between
            if any(g is not None for g in grads):
                raise RuntimeError("scade_amd: the gradient sink of this network was detached between the forward and "
                                   "the backward of a step")
            return (None,) * (5 + ctx.n_params)
        return (None, None, None, None, None) + tuple(grads)
