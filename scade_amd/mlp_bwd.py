"""Backward of the fused NeRF MLP: binds scade_mlp_bwd (dgrad chain + wgrad + reduce)."""
from __future__ import annotations

import math

from . import ops


def mlp_backward(net, acts, g_out):
    """-> list of 24 gradient tensors in ops.PARAM_ORDER (views of one flat buffer)."""
    if acts is None or acts.numel() == 0:
        raise RuntimeError("scade_amd: MLP backward called but the forward did not save activations")
    sink = getattr(net, "_grad_sink", None)
    if sink is not None:
        # FlatParams.attach_grad_sinks(): the 24 .grad tensors are consecutive views of one flat
        # buffer in PARAM_ORDER
        first = net.ordered_params()[0].grad
        if first is None or first.data_ptr() != sink.data_ptr() or sink.numel() != ops.N_PARAM_FLOATS:
            raise RuntimeError("scade_amd: stale gradient sink (parameters were re-homed after "
                               "FlatParams.attach_grad_sinks); call it again")
    # FlatParams.begin_step() declared the sink's contents dead: the FIRST backward of the step writes its
    # gradient straight into it (no zero fill before, no temporary, no add after); later ones accumulate
    direct = sink is not None and getattr(net, "_sink_fresh", False)
    out = sink if direct else None
    if net.train_precision in ("f16x3", "f16x3-dgrad"):
        flat = ops.mlp_bwd_f16(net.packed(), net.packed_t_f16(), acts, g_out,
                               wgrad_f16=net.train_precision == "f16x3", out=out)
    elif net.train_precision in ("f16", "bf16"):
        bf16 = net.train_precision == "bf16"
        flat = ops.mlp_bwd_lp(None, net.packed_t_lp(bf16), bf16, acts, g_out, out=out)
    elif net.train_precision == "f32":
        flat = ops.mlp_bwd(net.packed(), net.packed_t(), acts, g_out, out=out)
    else:
        raise ValueError("NeRF.train_precision must be one of %s" % (net.TRAIN_PRECISIONS,))
    if sink is not None:
        if direct:
            net._sink_fresh = False
        else:
            sink.add_(flat)      # ONE add replaces autograd's 24 per-tensor accumulations
        return [None] * len(ops.PARAM_ORDER)
    grads, o = [], 0
    for name in ops.PARAM_ORDER:
        shape = ops.PARAM_SHAPES[name]
        n = math.prod(shape)
        grads.append(flat[o:o + n].view(shape))
        o += n
    return grads
