"""Backward of the fused NeRF MLP (dgrad chain + wgrad) -- binds scade_mlp_bwd."""
from __future__ import annotations


def mlp_backward(net, mode, inp, viewdirs, bb, acts, g_out):
    raise NotImplementedError(
        "scade_amd: the MLP backward kernels (scade_mlp_bwd) are not built yet; "
        "there is no PyTorch fallback")
