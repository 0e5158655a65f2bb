"""Backward of the fused NeRF MLP: binds scade_mlp_bwd (dgrad chain + wgrad + reduce)."""
from __future__ import annotations

import math

from . import ops


class DeferredBackward:
    """Joins the MLP backward of the coarse and the fine NeRF of a train step into ONE dgrad launch, ONE
    weight-gradient launch and ONE reduce (scade_mlp_bwd2 / scade_mlp_bwd_lp2): their backward chains are
    independent (run_scade_scannet.py:711 detaches the samples between the two networks), and the joint grid
    fills whole rounds of two workgroups per CU where two separate launches each end in a part-filled one -
    which is most of the time at the 128 rays per GPU of a strongly scaled batch.

        with DeferredBackward() as q:        # inside: a network's FIRST backward of the step (the one that
            loss.backward(...)               # writes its gradient sink) only queues (net, acts, g_out)
        # on exit: two queued entries of one precision -> the pair launch; anything else -> one by one

    Only backwards that write a fresh gradient sink are deferred (they return no gradients to autograd, so
    nothing downstream waits for them); everything is launched on the stream current at exit."""
    active = None

    def __init__(self, after_net=None, defer_reduce=False):
        """``after_net(net)``: called when the gradient of the FIRST network of a joint launch (the one with fewer
        points: the coarse NeRF) is complete on the stream and the second one's weight gradient has not been launched
        yet - where a sharded step starts that network's share of the gradient exchange (Trainer(allreduce="staged"))."""
        self.items = []
        self.after_net = after_net
        # ``defer_reduce``: a joint launch leaves its partial rows unsummed; ``self.reduce`` = (ops.ReduceDesc, (first
        # network, second network) of the call) for ops.step_finish, or None when the flush took another route
        self.defer_reduce = bool(defer_reduce) and after_net is None
        self.reduce = None

    def __enter__(self):
        if DeferredBackward.active is not None:
            raise RuntimeError("DeferredBackward: already active")
        DeferredBackward.active = self
        return self

    def __exit__(self, et, ev, tb):
        DeferredBackward.active = None
        items, self.items = self.items, []
        if et is None:
            self.reduce = flush_deferred(items, self.after_net, self.defer_reduce)
        return False


def _pair_key(net):
    p = net.train_precision
    return p if (p == "f32" or p == "f16x3" or p in ops.LP_FORMATS) else None


def flush_deferred(items, after_net=None, defer=False):
    """-> (ops.ReduceDesc, (net of entry 0, net of entry 1)) when ``defer`` and the joint launch took it, else None"""
    if len(items) == 2 and _pair_key(items[0][0]) is not None and _pair_key(items[0][0]) == _pair_key(items[1][0]) \
            and items[0][0] is not items[1][0]:
        if after_net is not None:
            items = sorted(items, key=lambda it: it[2].numel())          # the shorter backward first
        (n0, a0, g0), (n1, a1, g1) = items
        prec = n0.train_precision
        if prec == "f16x3":
            if after_net is None:
                # (the larger call first: its workgroups are the launch's body, the smaller one's fill the tail)
                (n0, a0, g0), (n1, a1, g1) = sorted(items, key=lambda it: -it[2].numel())
                desc = ops.mlp_bwd_f16_2([n0.packed(), n1.packed()], [n0.packed_t_f16(), n1.packed_t_f16()], [a0, a1],
                                         [g0, g1], [n0._grad_sink, n1._grad_sink], defer=defer)
                return (desc, (n0, n1)) if desc is not None else None
            # (a staged gradient exchange wants the first network's gradient early: one launch sequence per network,
            # the shorter backward first - ``items`` was sorted above)
            for net, acts, g in items:
                _backward_now(net, acts, g, net._grad_sink)
                after_net(net)
            return
        sinks = [n0._grad_sink, n1._grad_sink]
        hook = None if after_net is None else (lambda: after_net(n0))
        if prec == "f32":
            desc = ops.mlp_bwd2([n0.packed(), n1.packed()], [n0.packed_t(), n1.packed_t()], [a0, a1], [g0, g1], sinks,
                                after_first=hook, defer=defer and hook is None)
            if after_net is not None:
                after_net(n1)
            return (desc, (n0, n1)) if desc is not None else None
        code, bf16 = ops.LP_FORMATS[prec]
        P0, P1 = g0.numel() // 4, g1.numel() // 4
        if ops.lp_point_tiles(P0) == ops.lp_point_tiles(P1):
            gm = [ops.GMAX_READY.pop(g.data_ptr(), None) for g in (g0, g1)]
            gmax = gm if (hook is None and all(t is not None for t in gm)) else None
            desc = ops.mlp_bwd_lp2([n0.packed_t_lp(bf16), n1.packed_t_lp(bf16)], code, [a0, a1], [g0, g1], sinks,
                                   after_first=hook, defer=defer and hook is None, gmax=gmax)
            if after_net is not None:
                after_net(n1)
            return (desc, (n0, n1)) if desc is not None else None
    for net, acts, g in items:
        _backward_now(net, acts, g, net._grad_sink)
        if after_net is not None:
            after_net(net)
    return None


def _backward_now(net, acts, g_out, out):
    if net.train_precision in ("f16x3", "f16x3-dgrad"):
        return ops.mlp_bwd_f16(net.packed(), net.packed_t_f16(), acts, g_out,
                               wgrad_f16=net.train_precision == "f16x3", out=out)
    if net.train_precision in ops.LP_FORMATS:
        code, bf16 = ops.LP_FORMATS[net.train_precision]
        return ops.mlp_bwd_lp(None, net.packed_t_lp(bf16), code, acts, g_out, out=out)
    if net.train_precision == "f32":
        return ops.mlp_bwd(net.packed(), net.packed_t(), acts, g_out, out=out)
    raise ValueError("NeRF.train_precision must be one of %s" % (net.TRAIN_PRECISIONS,))


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
    q = DeferredBackward.active
    if not direct and q is not None and any(it[0] is net for it in q.items):
        # the queued FIRST backward of this network OVERWRITES the sink when the queue is flushed: a contribution
        # added now would be lost (and a staged exchange would send the piece without it).  The SCADE step evaluates
        # each network once; a caller that evaluates one twice must not defer.
        raise RuntimeError("scade_amd: a second backward of a network whose first backward of the step is still "
                           "queued (DeferredBackward / Trainer(joint_backward=True)); use joint_backward=False")
    if direct and q is not None:
        # joined with the other network's backward when the queue is flushed (DeferredBackward.__exit__)
        q.items.append((net, acts, g_out))
        net._sink_fresh = False
        return [None] * len(ops.PARAM_ORDER)
    flat = _backward_now(net, acts, g_out, out)
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
