"""Tensor-level wrappers of the C ABI (include/scade_hip.h) and the autograd glue.

Everything numeric happens inside libscade_hip.so; this file only validates
tensors, allocates outputs with torch and records what the backward kernels need.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import call, check, check_current_device, last_error, load, ptr, stream

Tensor = torch.Tensor

PARAM_ORDER = ([f"pts_linears.{i}.{k}" for i in range(8) for k in ("weight", "bias")]
               + ["views_linears.0.weight", "views_linears.0.bias",
                  "feature_linear.weight", "feature_linear.bias",
                  "alpha_linear.weight", "alpha_linear.bias",
                  "rgb_linear.weight", "rgb_linear.bias"])

PARAM_SHAPES = {}
for _i in range(8):
    _k = 57 if _i == 0 else (313 if _i == 5 else 256)
    PARAM_SHAPES[f"pts_linears.{_i}.weight"] = (256, _k)
    PARAM_SHAPES[f"pts_linears.{_i}.bias"] = (256,)
PARAM_SHAPES.update({
    "views_linears.0.weight": (128, 259), "views_linears.0.bias": (128,),
    "feature_linear.weight": (256, 256), "feature_linear.bias": (256,),
    "alpha_linear.weight": (1, 256), "alpha_linear.bias": (1,),
    "rgb_linear.weight": (3, 128), "rgb_linear.bias": (3,),
})


class KernelTimer:
    """HIP-event stopwatch around individual kernel launches on the launch stream
    (torch.cuda.Event records on torch's current stream, which is the stream every
    scade_* call is enqueued on).  Used by bench.py for the roofline line."""

    def __init__(self):
        self.records = []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, e0, work):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((name, e0, e1, work))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, work in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out

    def by_work(self, name):
        """The records of one kernel split by launch size: {work per launch: {"launches", "ms"}} (one row
        per launch size, like rocprofv3's per-dispatch trace grouped by grid)."""
        torch.cuda.synchronize()
        out = {}
        for n, e0, e1, work in self.records:
            if n == name:
                d = out.setdefault(work, {"launches": 0, "ms": 0.0})
                d["launches"] += 1
                d["ms"] += e0.elapsed_time(e1)
        return out


KERNEL_TIMER: Optional[KernelTimer] = None
PARAM_EPOCH = 0   # bumped by optimizers that update parameters through raw pointers
N_PARAM_FLOATS = 589700
# 16-bit training formats: name -> (format code of the scade_mlp_*_lp entries, weight packs are bf16?)
#   "bf16-s8": bf16 arithmetic, the rows saved for the weight gradient (activations, dZ) stored as 8-bit e5m2
LP_FORMATS = {"f16": (0, False), "bf16": (1, True), "bf16-s8": (2, True)}
MLP_FLOP_PER_POINT = 2 * 587264      # algorithmic, unpadded


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: Tensor, what: str) -> Tuple[Tensor, int]:
    """2-D tensor whose rows are unit-stride -> (tensor, row stride in elements)."""
    check(t, what)
    if t.dim() != 2:
        raise ValueError(f"{what}: expected 2-D, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        # expanded (stride 0) rows are fine for read-only inputs handled by the callers
        if t.stride(0) != 0:
            t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1])


# ---------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------

def mlp_packed_floats() -> int:
    return int(_lib.load().scade_mlp_packed_floats())


def mlp_pack(params: Sequence[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """params: the 24 tensors in PARAM_ORDER (device, fp32, contiguous)."""
    if len(params) != 24:
        raise ValueError("mlp_pack: expected 24 parameter tensors")
    keep = []
    for name, p in zip(PARAM_ORDER, params):
        check(p, f"mlp_pack[{name}]")
        if tuple(p.shape) != PARAM_SHAPES[name]:
            raise ValueError(f"mlp_pack[{name}]: shape {tuple(p.shape)} != {PARAM_SHAPES[name]} "
                             "(only NeRF(D=8,W=256,input_ch=57,input_ch_views=3,skips=[4],"
                             "use_viewdirs=True) is implemented)")
        keep.append(_c(p.detach()))
    if out is None:
        out = torch.empty(mlp_packed_floats(), device=keep[0].device, dtype=torch.float32)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack", ctypes.cast(arr, ctypes.c_void_p), ptr(out), stream())
    return out


def mlp_pack_t(params: Sequence[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """Transposed weight pack for the dgrad chain (scade_mlp_pack_t)."""
    keep = [_c(check(p, "mlp_pack_t").detach()) for p in params]
    if out is None:
        out = torch.empty(int(_lib.load().scade_mlp_packed_t_floats()), device=keep[0].device,
                          dtype=torch.float32)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack_t", ctypes.cast(arr, ctypes.c_void_p), ptr(out), stream())
    return out


def mlp_pack_step(nets, fmt: str) -> None:
    """Bring the training weight packs of one or two NeRFs up to date with ONE launch (scade_mlp_pack_step): the
    forward and the transposed layout of each network whose cached blobs are stale.  ``fmt``: "f32" (the exact
    kernels' packs), "bf16" or "f16" (the 16-bit kernels' packs).  The blobs land in the networks' own caches
    (NeRF.packed / packed_t / packed_lp / packed_t_lp return them without packing again)."""
    lib = _lib.load()
    code = {"f32": 0, "bf16": 1, "f16": 2}[fmt]
    plist, fwd, tr, adopt = [], [], [], []
    for net in nets:
        ps = net.ordered_params()
        dev = ps[0].device
        key = net.pack_key()
        need_f, need_t = net.pack_stale(fmt, False, key), net.pack_stale(fmt, True, key)
        if not (need_f or need_t):
            continue
        # the 24 device pointers are part of ``key``; the tensor checks run once per set of pointers (they
        # cost 30 us per network and step otherwise), the parameters themselves keep the storage alive
        ptrs = tuple(k[0] for k in key[1:])
        if net.__dict__.get("_pack_checked") != ptrs:
            for p in ps:
                check(p, "mlp_pack_step")
            net.__dict__["_pack_checked"] = ptrs
        # the layout is NOT implied by the pointer (``w.data = w.data.t()`` keeps it): checked every time (~2 us)
        if not all(p.is_contiguous() for p in ps):
            raise ValueError("mlp_pack_step: parameters must be contiguous")
        # a network's blobs are allocated once per format and re-packed IN PLACE ever after: their addresses are
        # baked into captured steps (graphs.py), and step_finish() packs into them at the end of every step
        bf, bt = _train_blobs(net, fmt, dev, need_f, need_t)
        plist += ptrs
        fwd.append(bf)
        tr.append(bt)
        adopt.append((net, bf, bt, key))
    if not adopt:
        return
    vp = lambda ts: ctypes.cast((ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts]),
                                ctypes.c_void_p)
    call("scade_mlp_pack_step", len(adopt), ctypes.cast((ctypes.c_void_p * len(plist))(*plist), ctypes.c_void_p), code,
         vp(fwd), vp(tr), stream())
    for net, bf, bt, key in adopt:
        net.adopt_packs(fmt, bf, bt, key)


def _blob(cur, numel, dtype, dev):
    if cur is not None and cur.device == dev and cur.numel() == numel and cur.dtype == dtype:
        return cur
    return torch.empty(numel, device=dev, dtype=dtype)


def _train_blobs(net, fmt, dev, need_f=True, need_t=True):
    """(forward blob, transposed blob) of format "f32" | "bf16" | "f16" for ``net``: its existing tensors (to be
    re-packed in place) or fresh ones."""
    lib = _lib.load()
    d = net.__dict__
    if fmt == "f32":
        bf = _blob(d.get("_packed"), int(lib.scade_mlp_packed_floats()), torch.float32, dev) if need_f else None
        bt = _blob(d.get("_packed_t"), int(lib.scade_mlp_packed_t_floats()), torch.float32, dev) if need_t else None
    else:
        bf = _blob(d.get("_packed_lp"), int(lib.scade_mlp_packed_lp_bytes()), torch.uint8, dev) if need_f else None
        bt = _blob(d.get("_packed_t_lp"), int(lib.scade_mlp_packed_t_lp_bytes()), torch.uint8, dev) if need_t else None
    return bf, bt


def _train_blobs_f16x3(net, dev, need=(True, True, True)):
    lib = _lib.load()
    d = net.__dict__
    be = _blob(d.get("_packed"), int(lib.scade_mlp_packed_floats()), torch.float32, dev) if need[0] else None
    bf = _blob(d.get("_packed_f16"), int(lib.scade_mlp_packed_f16_bytes()), torch.uint8, dev) if need[1] else None
    bt = _blob(d.get("_packed_t_f16"), int(lib.scade_mlp_packed_t_f16_bytes()), torch.uint8, dev) if need[2] else None
    return be, bf, bt


def _adopt_f16x3(net, be, bf, bt, key):
    d = net.__dict__
    if be is not None:
        d["_packed"], d["_packed_key"] = be, key
    if bf is not None:
        d["_packed_f16"], d["_packed_f16_key"] = bf, key
    if bt is not None:
        d["_packed_t_f16"], d["_packed_t_f16_key"] = bt, key


# loss-scale maxima a step's loss launch produced for the output gradients it wrote: {g_out.data_ptr(): float32[256]
# device tensor}, consumed (popped) by the joint 16-bit backward of the same step (mlp_bwd.flush_deferred)
GMAX_READY = {}


class ReduceDesc:
    """What a deferred MLP backward leaves for ``step_finish``: the 64-byte descriptor scade_mlp_bwd*_deferred filled
    (partial-row pointers and row counts of the two entries of the call) and the workspaces it points into."""

    def __init__(self, keep):
        self.buf = ctypes.create_string_buffer(64)
        self.keep = keep

    def swap(self):
        """exchange the two entries (the descriptor is indexed by the network's position in the optimizer's segment)"""
        b = self.buf.raw
        self.buf.raw = b[8:16] + b[0:8] + b[20:24] + b[16:20] + b[40:56] + b[24:40] + b[56:64]
        return self


def step_finish(opt_a, opt_b, n_nets: int, lr_a=None, dev: bool = False, reduce: Optional[ReduceDesc] = None) -> None:
    """``optim.adam_step_pair`` with the weight gradient's reduce inside the launch (scade_step_finish): [``reduce``: the
    sum of a deferred backward's partial rows, written to the bucket on the way ->] ``opt_a.step(lr=lr_a)`` on the
    ``n_nets`` networks' segment and (``opt_b`` given) ``opt_b.step()`` on the scale / shift segment.  ``dev``: the
    optimizers' scalars come from their device states, already advanced for this step."""
    opts = [opt_a] + ([opt_b] if opt_b is not None else [])
    for o in opts:
        o.steps += 1
    two = lambda f, ct: (ct * 2)(*[f(o) for o in opts] + ([ct()] if len(opts) == 1 else []))
    P = ctypes.c_void_p
    pp = lambda f: ctypes.cast(two(lambda o: f(o).data_ptr(), ctypes.c_void_p), P)
    n = two(lambda o: o.flat.numel, ctypes.c_long)
    if opt_a.flat.numel != n_nets * N_PARAM_FLOATS:
        raise ValueError("step_finish: the first optimizer's segment must be the networks' parameters")
    if dev:
        scal = (None,) * 6 + (pp(lambda o: o.state),)
    else:
        scal = (ctypes.cast(two(lambda o: float(lr_a if (o is opt_a and lr_a is not None) else o.lr), ctypes.c_float), P),
                ctypes.cast(two(lambda o: float(o.betas[0]), ctypes.c_float), P),
                ctypes.cast(two(lambda o: float(o.betas[1]), ctypes.c_float), P),
                ctypes.cast(two(lambda o: float(o.eps), ctypes.c_float), P),
                ctypes.cast(two(lambda o: int(o.steps), ctypes.c_int), P),
                ctypes.cast(two(lambda o: 1.0, ctypes.c_float), P), None)
    call("scade_step_finish", pp(lambda o: o.flat.data), pp(lambda o: o.flat.grad), pp(lambda o: o.exp_avg),
         pp(lambda o: o.exp_avg_sq), ctypes.cast(n, P), *scal, None if reduce is None else ctypes.cast(reduce.buf, P),
         n_nets, stream())
    global PARAM_EPOCH
    PARAM_EPOCH += 1


def mlp_pack_step_f16x3(nets) -> None:
    """The same for the split-precision training kernels (scade_mlp_pack_step_f16x3): per network the exact forward
    blob (the dgrad's heads read the fp32 head weights), the two-plane forward blob and the two-plane transposed blob
    - six stand-alone launches per step otherwise.  The blobs land in NeRF.packed / packed_f16 / packed_t_f16's caches."""
    lib = _lib.load()
    plist, ex, fw, tr, adopt = [], [], [], [], []
    for net in nets:
        ps = net.ordered_params()
        dev = ps[0].device
        key = net.pack_key()
        d = net.__dict__
        stale = lambda blob, k: blob is None or k != key or blob.device != dev
        need = (stale(d.get("_packed"), d.get("_packed_key")), stale(d.get("_packed_f16"), d.get("_packed_f16_key")),
                stale(d.get("_packed_t_f16"), d.get("_packed_t_f16_key")))
        if not any(need):
            continue
        ptrs = tuple(k[0] for k in key[1:])
        if d.get("_pack_checked") != ptrs:
            for p in ps:
                check(p, "mlp_pack_step_f16x3")
            d["_pack_checked"] = ptrs
        if not all(p.is_contiguous() for p in ps):
            raise ValueError("mlp_pack_step_f16x3: parameters must be contiguous")
        be, bf, bt = _train_blobs_f16x3(net, dev, need)
        plist += ptrs
        ex.append(be)
        fw.append(bf)
        tr.append(bt)
        adopt.append((net, be, bf, bt, key))
    if not adopt:
        return
    vp = lambda ts: ctypes.cast((ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts]),
                                ctypes.c_void_p)
    call("scade_mlp_pack_step_f16x3", len(adopt), ctypes.cast((ctypes.c_void_p * len(plist))(*plist), ctypes.c_void_p),
         vp(ex), vp(fw), vp(tr), stream())
    for net, be, bf, bt, key in adopt:
        _adopt_f16x3(net, be, bf, bt, key)


def mlp_acts_alloc(P: int, device) -> Tensor:
    return torch.empty(int(_lib.load().scade_mlp_acts_floats(P)), device=device, dtype=torch.float32)


def _grad_out(out: Optional[Tensor], device) -> Tensor:
    """the flat gradient buffer of an MLP backward: a fresh tensor, or the caller's (OVERWRITTEN: the reduce
    kernel writes every one of its N_PARAM_FLOATS elements)"""
    if out is None:
        return torch.empty(N_PARAM_FLOATS, device=device, dtype=torch.float32)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != N_PARAM_FLOATS or out.device != device:
        raise ValueError("mlp_bwd: out must be a contiguous fp32 tensor of N_PARAM_FLOATS elements on the device of g_out")
    return out


def mlp_bwd(packed: Tensor, packed_t: Tensor, acts: Tensor, g_out: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """-> flat gradient [589700] in PARAM_ORDER (written into ``out`` when given)."""
    g = _c(check(g_out, "mlp_bwd: g_out")).reshape(-1, 4)
    P = g.shape[0]
    ws = torch.empty(int(_lib.load().scade_mlp_bwd_workspace_floats(P)), device=g.device,
                     dtype=torch.float32)
    grad = _grad_out(out, g.device)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_bwd", ptr(packed), ptr(packed_t), ptr(acts), ptr(g), P, ptr(ws), ptr(grad), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P) * 2 * MLP_FLOP_PER_POINT)
    return grad


def mlp_pack_f16(params: Sequence[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """Two-plane fp16 weight pack for the split-precision inference kernel."""
    keep = [_c(check(p, "mlp_pack_f16").detach()) for p in params]
    if out is None:
        out = torch.empty(int(_lib.load().scade_mlp_packed_f16_bytes()), device=keep[0].device,
                          dtype=torch.uint8)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack_f16", ctypes.cast(arr, ctypes.c_void_p), ptr(out), stream())
    return out


def mlp_pack_t_f16(params: Sequence[Tensor]) -> Tensor:
    keep = [_c(check(p, "mlp_pack_t_f16").detach()) for p in params]
    out = torch.empty(int(_lib.load().scade_mlp_packed_t_f16_bytes()), device=keep[0].device, dtype=torch.uint8)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack_t_f16", ctypes.cast(arr, ctypes.c_void_p), ptr(out), stream())
    return out


def mlp_bwd_f16(packed: Tensor, packed_t_f16: Tensor, acts: Tensor, g_out: Tensor,
                wgrad_f16: bool = True, out: Optional[Tensor] = None) -> Tensor:
    """Split-precision dgrad + weight gradient -> flat gradient [589700] in PARAM_ORDER.  ``wgrad_f16``: the
    split-precision weight gradient on 24-bit saved rows (the forward must have run with ``rows24=True``); False: the
    exact fp32 weight gradient on fp32 rows."""
    g = _c(check(g_out, "mlp_bwd_f16: g_out")).reshape(-1, 4)
    P = g.shape[0]
    ws = torch.empty(int(_lib.load().scade_mlp_bwd_workspace_floats(P)), device=g.device, dtype=torch.float32)
    grad = _grad_out(out, g.device)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_bwd_f16", ptr(packed), ptr(packed_t_f16), ptr(acts), ptr(g), P, int(wgrad_f16), ptr(ws),
         ptr(grad), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P) * 2 * MLP_FLOP_PER_POINT)
    return grad


def mlp_bwd_f16_2(packed, packed_t_f16, acts, g_out, outs, defer: bool = False):
    """Split-precision backward (24-bit saved rows) of TWO network calls - the coarse + fine NeRF of a train step - as
    one zeroing launch, one dgrad launch, one weight-gradient launch and one reduce (scade_mlp_bwd_f16_2).  Every
    argument: a pair; ``outs`` = the two flat gradient buffers [589700], OVERWRITTEN.  Same bits as two
    ``mlp_bwd_f16`` calls (same tiles, same chunks, same summation order).  ``defer``: no reduce launch - the partial
    rows stay in the workspaces and the returned ``ReduceDesc`` hands them to ``step_finish`` (``outs`` untouched)."""
    g = [_c(check(t, "mlp_bwd_f16_2: g_out")).reshape(-1, 4) for t in g_out]
    P = [t.shape[0] for t in g]
    lib = _lib.load()
    ws = [torch.empty(int(lib.scade_mlp_bwd_workspace_floats(P[i])), device=g[i].device, dtype=torch.float32)
          for i in range(2)]
    grads = [_grad_out(o, g[0].device) for o in outs]
    Pa = (ctypes.c_int * 2)(*P)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    desc = None
    if defer:
        desc = ReduceDesc(ws)
        call("scade_mlp_bwd_f16_2_deferred", _host_ptrs(packed), _host_ptrs(packed_t_f16), _host_ptrs(acts), _host_ptrs(g),
             ctypes.cast(Pa, ctypes.c_void_p), _host_ptrs(ws), ctypes.cast(desc.buf, ctypes.c_void_p), stream())
    else:
        call("scade_mlp_bwd_f16_2", _host_ptrs(packed), _host_ptrs(packed_t_f16), _host_ptrs(acts), _host_ptrs(g),
             ctypes.cast(Pa, ctypes.c_void_p), 1, _host_ptrs(ws), _host_ptrs(grads), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P[0] + P[1]) * 2 * MLP_FLOP_PER_POINT)
    return desc


def mlp_fwd_f16(packed_f16: Tensor, inp: Tensor, viewdirs: Optional[Tensor], bb: Optional[Tensor],
                acts: Optional[Tensor] = None, rows24: bool = False) -> Tensor:
    """Split-precision forward: inp [P,60] (viewdirs None) or pts [N,S,3] + viewdirs [N,3] + bb [4].  ``rows24``
    (with ``acts``): the saved rows in the 24-bit form the split-precision weight gradient reads (mode + 2) - what
    ``mlp_bwd_f16(wgrad_f16=True)`` expects; False: fp32 rows (``wgrad_f16=False``, the exact weight gradient)."""
    check(inp, "mlp_fwd_f16: input")
    inp = _c(inp)
    if rows24 and acts is None:
        raise ValueError("mlp_fwd_f16: rows24 needs the training workspace")
    fmt = 2 if rows24 else 0
    if viewdirs is None:
        if inp.dim() != 2 or inp.shape[1] != 60:
            raise ValueError("mlp_fwd_f16: x must be [P,60]")
        P = inp.shape[0]
        out = torch.empty(P, 4, device=inp.device, dtype=torch.float32)
        call("scade_mlp_fwd_f16", ptr(packed_f16), 0 + fmt, ptr(inp), None, 0, None, P, 1, ptr(out), ptr(acts), stream())
        return out
    N, S = inp.shape[0], inp.shape[1]
    viewdirs, vstride = _rows(viewdirs, "mlp_fwd_f16: viewdirs")
    bb = _c(check(bb, "mlp_fwd_f16: bb"))
    out = torch.empty(N, S, 4, device=inp.device, dtype=torch.float32)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_fwd_f16", ptr(packed_f16), 1 + fmt, ptr(inp), ptr(viewdirs), vstride, ptr(bb), N * S, S,
         ptr(out), ptr(acts), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_fwd_f16_kernel", t0, float(N * S) * MLP_FLOP_PER_POINT)
    return out


def mlp_pack_lp(params: Sequence[Tensor], bf16: bool) -> Tensor:
    """Single-plane 16-bit weight pack (fp16, or bf16 when ``bf16``) for scade_mlp_fwd_lp."""
    keep = [_c(check(p, "mlp_pack_lp").detach()) for p in params]
    out = torch.empty(int(_lib.load().scade_mlp_packed_lp_bytes()), device=keep[0].device, dtype=torch.uint8)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack_lp", ctypes.cast(arr, ctypes.c_void_p), ptr(out), int(bf16), stream())
    return out


def mlp_acts_lp_alloc(P: int, device) -> Tensor:
    return torch.empty(int(_lib.load().scade_mlp_acts_lp_bytes(P)), device=device, dtype=torch.uint8)


def mlp_pack_t_lp(params: Sequence[Tensor], bf16: bool) -> Tensor:
    keep = [_c(check(p, "mlp_pack_t_lp").detach()) for p in params]
    out = torch.empty(int(_lib.load().scade_mlp_packed_t_lp_bytes()), device=keep[0].device, dtype=torch.uint8)
    arr = (ctypes.c_void_p * 24)(*[t.data_ptr() for t in keep])
    call("scade_mlp_pack_t_lp", ctypes.cast(arr, ctypes.c_void_p), ptr(out), int(bf16), stream())
    return out


def mlp_bwd_lp(packed: Optional[Tensor], packed_t_lp: Tensor, bf16: bool, acts: Tensor, g_out: Tensor,
               out: Optional[Tensor] = None) -> Tensor:
    """16-bit dgrad + wgrad (fp32 accumulate, power-of-two loss scaling) -> flat gradient [589700]."""
    g = _c(check(g_out, "mlp_bwd_lp: g_out")).reshape(-1, 4)
    P = g.shape[0]
    ws = torch.empty(int(_lib.load().scade_mlp_bwd_lp_workspace_bytes(P)), device=g.device, dtype=torch.uint8)
    grad = _grad_out(out, g.device)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_bwd_lp", ptr(packed), ptr(packed_t_lp), int(bf16), ptr(acts), ptr(g), P, ptr(ws), ptr(grad),
         stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P) * 2 * MLP_FLOP_PER_POINT)
    return grad


def _host_ptrs(ts):
    """HOST array of device pointers (the scade_*2 entries take two-entry host arrays)."""
    return ctypes.cast((ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), ctypes.c_void_p)


def mlp_bwd2(packed, packed_t, acts, g_out, outs, after_first=None, defer: bool = False):
    """Exact backward of TWO network calls (coarse + fine NeRF of a train step) as one dgrad launch, one
    weight-gradient launch and one reduce (scade_mlp_bwd2).  Every argument: a pair; ``outs`` = the two flat
    gradient buffers [589700], OVERWRITTEN.  ``after_first``: a callable run when the FIRST entry's gradient is
    complete on the stream (scade_mlp_bwd2_phases: joint dgrad, entry 0's weight gradient + reduce, callback, entry
    1's) - a sharded step starts entry 0's gradient exchange there, under entry 1's weight gradient.  ``defer``: no
    reduce launch - the returned ``ReduceDesc`` hands the partial rows to ``step_finish`` (``outs`` untouched)."""
    g = [_c(check(t, "mlp_bwd2: g_out")).reshape(-1, 4) for t in g_out]
    P = [t.shape[0] for t in g]
    lib = _lib.load()
    ws = [torch.empty(int(lib.scade_mlp_bwd2_workspace_floats(P[i], P[1 - i])), device=g[i].device, dtype=torch.float32)
          for i in range(2)]
    grads = [_grad_out(o, g[0].device) for o in outs]
    Pa = (ctypes.c_int * 2)(*P)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    args = (_host_ptrs(packed), _host_ptrs(packed_t), _host_ptrs(acts), _host_ptrs(g),
            ctypes.cast(Pa, ctypes.c_void_p), _host_ptrs(ws), _host_ptrs(grads))
    desc = None
    if defer:
        if after_first is not None:
            raise ValueError("mlp_bwd2: a deferred reduce has no first-network hook")
        desc = ReduceDesc(ws)
        call("scade_mlp_bwd2_deferred", *args[:6], ctypes.cast(desc.buf, ctypes.c_void_p), stream())
    elif after_first is None:
        call("scade_mlp_bwd2", *args, stream())
    else:
        call("scade_mlp_bwd2_phases", *args, 3, stream())
        after_first()
        call("scade_mlp_bwd2_phases", *args, 4, stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P[0] + P[1]) * 2 * MLP_FLOP_PER_POINT)
    return desc


def mlp_bwd_lp2(packed_t_lp, bf16: bool, acts, g_out, outs, after_first=None, defer: bool = False, gmax=None):
    """16-bit backward of two network calls in one launch each (scade_mlp_bwd_lp2); see mlp_bwd2.  ``gmax``: per entry
    the 256 loss-scale maxima the step's loss launch already produced (``GMAX_READY``), or None."""
    g = [_c(check(t, "mlp_bwd_lp2: g_out")).reshape(-1, 4) for t in g_out]
    P = [t.shape[0] for t in g]
    lib = _lib.load()
    ws = [torch.empty(int(lib.scade_mlp_bwd_lp2_workspace_bytes(P[i], P[1 - i])), device=g[i].device, dtype=torch.uint8)
          for i in range(2)]
    grads = [_grad_out(o, g[0].device) for o in outs]
    Pa = (ctypes.c_int * 2)(*P)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    args = (_host_ptrs(packed_t_lp), int(bf16), _host_ptrs(acts), _host_ptrs(g),
            ctypes.cast(Pa, ctypes.c_void_p), _host_ptrs(ws), _host_ptrs(grads))
    desc = None
    if defer or gmax is not None:
        if after_first is not None:
            raise ValueError("mlp_bwd_lp2: a deferred reduce / precomputed maxima have no first-network hook")
        desc = ReduceDesc(ws) if defer else None
        gm = None if gmax is None else ctypes.cast((ctypes.c_void_p * 2)(*[None if t is None else t.data_ptr() for t in gmax]),
                                                   ctypes.c_void_p)
        call("scade_mlp_bwd_lp2_deferred", *args, gm, None if desc is None else ctypes.cast(desc.buf, ctypes.c_void_p),
             stream())
    elif after_first is None:
        call("scade_mlp_bwd_lp2", *args, stream())
    else:
        call("scade_mlp_bwd_lp2_phases", *args, 3, stream())
        after_first()
        call("scade_mlp_bwd_lp2_phases", *args, 4, stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_bwd", t0, float(P[0] + P[1]) * 2 * MLP_FLOP_PER_POINT)
    return desc


class CoarsePoints:
    """The static coarse-sample buffers of a graph-captured step and what the launch in front of it needs to fill them
    (scade_stage_inputs_points / scade_gather_batch_points = ray_points_draw inside that launch): ``z`` [N,S], ``pts``
    [N,S,3], ``u_a`` / ``u_b`` [N,Si] (None: that sampler's draws are injected); ``key()`` -> the Philox key, ``step()``
    -> the host's count of optimizer steps taken (the step index the captured kernel would read from the device)."""

    def __init__(self, n_rays, n_samples, n_importance, lindisp, device, key, step, want_a=True, want_b=True):
        self.N, self.S, self.Si, self.lindisp = int(n_rays), int(n_samples), int(n_importance), bool(lindisp)
        self.z = torch.empty(n_rays, n_samples, device=device, dtype=torch.float32)
        self.pts = torch.empty(n_rays, n_samples, 3, device=device, dtype=torch.float32)
        self.u_a = torch.empty(n_rays, n_importance, device=device, dtype=torch.float32) if want_a else None
        self.u_b = torch.empty(n_rays, n_importance, device=device, dtype=torch.float32) if want_b else None
        self.t_vals = linspace01(n_samples, device)
        self.key, self.step = key, step

    def tail_args(self):
        return (ptr(self.t_vals), self.S, int(self.lindisp), int(self.key()) & (2 ** 64 - 1), int(self.step()), self.Si,
                ptr(self.z), ptr(self.pts), ptr(self.u_a), ptr(self.u_b))


class StepPacks:
    """The weight packs of a graph-captured step done by the launch in front of it (``stage_inputs`` / ``ResidentBatchGather``
    ``packs=``): the training blobs of format ``fmt`` ("f32" | "bf16" | "f16" | "f16x3") of ``nets`` are re-packed IN
    PLACE from the current parameters.  ``prepare()`` allocates missing blobs, packs them once and marks them fresh
    (call it right before a capture: the captured body's own pack then finds nothing to do)."""

    def __init__(self, nets, fmt: str):
        self.nets, self.fmt = list(nets), fmt
        self.code = {"f32": 0, "bf16": 1, "f16": 2, "f16x3": 3}[fmt]

    def prepare(self):
        (mlp_pack_step_f16x3 if self.fmt == "f16x3" else (lambda nets: mlp_pack_step(nets, self.fmt)))(self.nets)

    def tail_args(self):
        plist, ex, fw, tr = [], [], [], []
        for net in self.nets:
            d = net.__dict__
            plist += [p.data_ptr() for p in net.ordered_params()]
            if self.code == 3:
                be, bf, bt = d.get("_packed"), d.get("_packed_f16"), d.get("_packed_t_f16")
            elif self.code == 0:
                be, bf, bt = d.get("_packed"), None, d.get("_packed_t")
            else:
                be, bf, bt = None, d.get("_packed_lp"), d.get("_packed_t_lp")
            if (be is None and bf is None) or bt is None:
                raise RuntimeError("StepPacks: the networks' training blobs do not exist yet (prepare() first)")
            ex.append(be); fw.append(bf); tr.append(bt)
        P = ctypes.c_void_p
        vp = lambda ts: ctypes.cast((P * len(ts))(*[None if t is None else t.data_ptr() for t in ts]), P)
        return (self.code, len(self.nets), ctypes.cast((P * len(plist))(*plist), P), vp(ex), vp(fw), vp(tr))


_NO_PACK = (-1, 0, None, None, None, None)


def stage_inputs(pairs, scalar=None, tick=None, points=None, rays=None, packs=None) -> None:
    """``dst.copy_(src)`` for up to eight (src, dst) pairs - and ``scalar = (int64 tensor, value)``: one 8-byte
    store; ``tick = [state, state | None]``: the device-resident scalars of up to two fused optimizers advanced by
    one step - in ONE launch (scade_stage_inputs): the prologue of a graph-captured step.  Pairs that the
    kernel does not take as they are (other dtype / shape / layout / device) fall back to ``copy_``.  ``points`` +
    ``rays``: see CoarsePoints."""
    src_p, dst_p, nbytes = [], [], []
    for src, dst in pairs:
        if src.data_ptr() == dst.data_ptr() and src.shape == dst.shape:
            continue                                  # the caller filled the static buffer itself
        ok = (src.is_cuda and src.device == dst.device and src.dtype == dst.dtype and src.shape == dst.shape
              and src.is_contiguous() and dst.is_contiguous() and (src.numel() * src.element_size()) % 4 == 0
              and src.data_ptr() % 4 == 0 and dst.data_ptr() % 4 == 0 and len(src_p) < 8)
        if not ok:
            dst.copy_(src)
            continue
        src_p.append(src.data_ptr()); dst_p.append(dst.data_ptr()); nbytes.append(src.numel() * src.element_size())
    sd, sv = (scalar[0].data_ptr(), int(scalar[1])) if scalar is not None else (None, 0)
    if scalar is not None and (scalar[0].dtype != torch.int64 or not scalar[0].is_cuda):
        raise ValueError("stage_inputs: the scalar destination must be an int64 device tensor")
    ticks = None
    if tick is not None and any(t is not None for t in tick):
        for t in tick:
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() >= 16 and t.is_contiguous()):
                raise ValueError("stage_inputs: tick entries are the float32[16] device states of FusedAdam")
        t2 = (list(tick) + [None])[:2]
        ticks = ctypes.cast((ctypes.c_void_p * 2)(*[None if t is None else t.data_ptr() for t in t2]), ctypes.c_void_p)
    if not src_p and sd is None and ticks is None and points is None and packs is None:
        return
    n = len(src_p)
    vp = lambda v: ctypes.cast((ctypes.c_void_p * max(n, 1))(*v), ctypes.c_void_p)
    if points is not None or packs is not None:
        # ``points`` (a CoarsePoints) + ``rays`` (the SOURCE ray rows of this step): ray_points_draw rides in the launch;
        # ``packs`` (a StepPacks): so do the step's weight packs
        pk = _NO_PACK if packs is None else packs.tail_args()
        if points is not None:
            rr, stride = _rows(rays, "stage_inputs: rays")
            if rr.shape[0] != points.N or rr.shape[1] < 8:
                raise ValueError("stage_inputs: rays must be the step's [N, >= 8] ray rows")
            t = points.tail_args()
            pa = (ptr(rr), stride, t[0], points.N) + tuple(t[1:])
        else:
            pa = (None, 8, None, 0, 1, 0, 0, 0, 0, None, None, None, None)
        call("scade_stage_inputs_points", vp(src_p), vp(dst_p),
             ctypes.cast((ctypes.c_long * max(n, 1))(*nbytes), ctypes.c_void_p), n, sd, sv, ticks, *pa, *pk, stream())
        return
    call("scade_stage_inputs", vp(src_p), vp(dst_p), ctypes.cast((ctypes.c_long * max(n, 1))(*nbytes), ctypes.c_void_p), n,
         sd, sv, ticks, stream())


def lp_point_tiles(P: int) -> int:
    return int(_lib.load().scade_mlp_lp_point_tiles(int(P)))


def mlp_fwd_lp(packed_lp: Tensor, bf16: bool, inp: Tensor, viewdirs: Optional[Tensor],
               bb: Optional[Tensor], acts: Optional[Tensor] = None) -> Tensor:
    """16-bit-operand forward: inp [P,60] (viewdirs None) or pts [N,S,3] + viewdirs [N,3] + bb [4]."""
    check(inp, "mlp_fwd_lp: input")
    inp = _c(inp)
    if viewdirs is None:
        if inp.dim() != 2 or inp.shape[1] != 60:
            raise ValueError("mlp_fwd_lp: x must be [P,60]")
        P = inp.shape[0]
        out = torch.empty(P, 4, device=inp.device, dtype=torch.float32)
        call("scade_mlp_fwd_lp", ptr(packed_lp), int(bf16), 0, ptr(inp), None, 0, None, P, 1, ptr(out), ptr(acts),
             stream())
        return out
    N, S = inp.shape[0], inp.shape[1]
    viewdirs, vstride = _rows(viewdirs, "mlp_fwd_lp: viewdirs")
    bb = _c(check(bb, "mlp_fwd_lp: bb"))
    out = torch.empty(N, S, 4, device=inp.device, dtype=torch.float32)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_fwd_lp", ptr(packed_lp), int(bf16), 1, ptr(inp), ptr(viewdirs), vstride, ptr(bb), N * S, S,
         ptr(out), ptr(acts), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_fwd_lp_kernel", t0, float(N * S) * MLP_FLOP_PER_POINT)
    return out


def mlp_fwd_embedded(packed: Tensor, x: Tensor, acts: Optional[Tensor] = None) -> Tensor:
    """NeRF.forward on x[P,60] (mode 0)."""
    check(x, "mlp_fwd: x")
    if x.dim() != 2 or x.shape[1] != 60:
        raise ValueError(f"mlp_fwd: x must be [P,60], got {tuple(x.shape)}")
    x = _c(x)
    P = x.shape[0]
    out = torch.empty(P, 4, device=x.device, dtype=torch.float32)
    call("scade_mlp_fwd", ptr(packed), 0, ptr(x), None, 0, None, P, 1, ptr(out), ptr(acts), stream())
    return out


def mlp_fwd_points(packed: Tensor, pts: Tensor, viewdirs: Tensor, bb: Tensor,
                   acts: Optional[Tensor] = None) -> Tensor:
    """run_network fused (mode 1): pts [N,S,3], viewdirs [N,3], bb [4] -> raw [N,S,4]."""
    check(pts, "mlp_fwd: pts"); check(viewdirs, "mlp_fwd: viewdirs"); check(bb, "mlp_fwd: bb")
    if pts.dim() != 3 or pts.shape[-1] != 3:
        raise ValueError(f"mlp_fwd: pts must be [N,S,3], got {tuple(pts.shape)}")
    N, S = pts.shape[0], pts.shape[1]
    if tuple(viewdirs.shape) != (N, 3):
        raise ValueError(f"mlp_fwd: viewdirs must be [{N},3], got {tuple(viewdirs.shape)}")
    viewdirs, vstride = _rows(viewdirs, "mlp_fwd: viewdirs")
    pts, bb = _c(pts), _c(bb)
    out = torch.empty(N, S, 4, device=pts.device, dtype=torch.float32)
    t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
    call("scade_mlp_fwd", ptr(packed), 1, ptr(pts), ptr(viewdirs), vstride, ptr(bb), N * S, S,
         ptr(out), ptr(acts), stream())
    if t0 is not None:
        KERNEL_TIMER.stop("mlp_fwd_kernel", t0, float(N * S) * MLP_FLOP_PER_POINT)
    return out


def embed(x: Tensor, multires: int) -> Tensor:
    check(x, "embed: x")
    D = x.shape[-1]
    flat = _c(x.reshape(-1, D))
    out = torch.empty(flat.shape[0], D * (1 + 2 * multires), device=x.device, dtype=torch.float32)
    call("scade_embed", ptr(flat), flat.shape[0], D, multires, ptr(out), stream())
    return out.reshape(*x.shape[:-1], out.shape[-1])


# ---------------------------------------------------------------------------
# per-ray operators (raw, no autograd)
# ---------------------------------------------------------------------------

_LINSPACE_CACHE = {}


def linspace01(steps: int, device) -> Tensor:
    """torch.linspace(0,1,steps) computed on the HOST (bit-identical to the CPU
    reference) and cached on the device."""
    key = (steps, str(device))
    t = _LINSPACE_CACHE.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, steps=steps, device="cpu").to(device)
        _LINSPACE_CACHE[key] = t
    return t


def ray_points(rays: Tensor, n_samples: int, t_rand: Optional[Tensor], lindisp: bool,
               want_pts: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    rays, stride = _rows(rays, "ray_points: rays")
    N = rays.shape[0]
    if rays.shape[1] < 8:
        raise ValueError("ray_points: rays need >= 8 columns (o, d, near, far)")
    z = torch.empty(N, n_samples, device=rays.device, dtype=torch.float32)
    pts = torch.empty(N, n_samples, 3, device=rays.device, dtype=torch.float32) if want_pts else None
    if t_rand is not None:
        t_rand = _c(check(t_rand, "ray_points: t_rand"))
        if tuple(t_rand.shape) != (N, n_samples):
            raise ValueError("ray_points: t_rand must be [N,S]")
    call("scade_ray_points", ptr(rays), stride, ptr(linspace01(n_samples, rays.device)), ptr(t_rand),
         N, n_samples, int(bool(lindisp)), ptr(z), ptr(pts), stream())
    return z, pts


class Draws:
    """Where a training step's uniform draws come from when they are made inside scade_ray_points_draw:
    ``seed`` (Philox key), ``step`` (host step index) or ``step_dev`` (a device float holding the number of steps
    taken - FusedAdam.state[0:1] - for graph-captured steps, where nothing may change on the host)."""

    def __init__(self, seed: int, step: int = 0, step_dev: Optional[Tensor] = None):
        self.seed, self.step, self.step_dev = int(seed) & (2 ** 64 - 1), int(step), step_dev


def ray_points_draw(rays: Tensor, n_samples: int, lindisp: bool, draws: Draws, n_importance: int,
                    want_a: bool = True, want_b: bool = True):
    """ray_points with the stratified jitter drawn in the kernel -> (z, pts, u_a, u_b): u_a / u_b [N,n_importance]
    = the draws of the coarse importance sampler / the depth-hypothesis sampler (None when not wanted)."""
    rays, stride = _rows(rays, "ray_points_draw: rays")
    N = rays.shape[0]
    if rays.shape[1] < 8:
        raise ValueError("ray_points_draw: rays need >= 8 columns (o, d, near, far)")
    dev = rays.device
    z = torch.empty(N, n_samples, device=dev, dtype=torch.float32)
    pts = torch.empty(N, n_samples, 3, device=dev, dtype=torch.float32)
    u_a = torch.empty(N, n_importance, device=dev, dtype=torch.float32) if want_a else None
    u_b = torch.empty(N, n_importance, device=dev, dtype=torch.float32) if want_b else None
    if draws.step_dev is not None:
        check(draws.step_dev, "ray_points_draw: step_dev")
    call("scade_ray_points_draw", ptr(rays), stride, ptr(linspace01(n_samples, dev)), N, n_samples,
         int(bool(lindisp)), draws.seed, draws.step, ptr(draws.step_dev), n_importance, ptr(z), ptr(pts), ptr(u_a),
         ptr(u_b), stream())
    return z, pts, u_a, u_b


def perturb_z(z_vals: Tensor, t_rand: Tensor) -> Tensor:
    z = _c(check(z_vals, "perturb_z_vals: z_vals"))
    t = _c(check(t_rand, "perturb_z_vals: t_rand"))
    if z.shape != t.shape:
        raise ValueError("perturb_z_vals: z_vals and t_rand shapes differ")
    S = z.shape[-1]
    out = torch.empty_like(z)
    call("scade_perturb_z", ptr(z), ptr(t), z.numel() // S, S, ptr(out), stream())
    return out


def composite_fwd(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise: Optional[Tensor] = None):
    check(raw, "raw2outputs: raw"); check(z_vals, "raw2outputs: z_vals")
    N, S = z_vals.shape
    if tuple(raw.shape) != (N, S, 4):
        raise ValueError(f"raw2outputs: raw must be [{N},{S},4], got {tuple(raw.shape)}")
    raw, z_vals = _c(raw), _c(z_vals)
    rays_d, ds = _rows(rays_d, "raw2outputs: rays_d")
    if noise is not None:
        noise = _c(check(noise, "raw2outputs: noise"))
    dev = raw.device
    rgb = torch.empty(N, 3, device=dev); disp = torch.empty(N, device=dev)
    acc = torch.empty(N, device=dev); w = torch.empty(N, S, device=dev); depth = torch.empty(N, device=dev)
    call("scade_composite_fwd", ptr(raw), ptr(z_vals), ptr(rays_d), ds, ptr(noise), N, S, ptr(rgb),
         ptr(disp), ptr(acc), ptr(w), ptr(depth), stream())
    return rgb, disp, acc, w, depth


def composite_bwd(raw, z_vals, rays_d, noise, g_rgb, g_disp, g_acc, g_w, g_depth) -> Tensor:
    N, S = z_vals.shape
    raw, z_vals = _c(raw), _c(z_vals)
    rays_d, ds = _rows(rays_d, "raw2outputs.backward: rays_d")
    gs = [None if g is None else _c(g) for g in (g_rgb, g_disp, g_acc, g_w, g_depth)]
    g_raw = torch.empty(N, S, 4, device=raw.device, dtype=torch.float32)
    call("scade_composite_bwd", ptr(raw), ptr(z_vals), ptr(rays_d), ds, ptr(noise), N, S,
         ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]), ptr(g_raw), stream())
    return g_raw


def _u_arg(u: Tensor, N: int, S: int) -> Tuple[Tensor, int]:
    check(u, "sample_pdf: u")
    if u.dim() == 1:
        if u.shape[0] != S:
            raise ValueError("sample_pdf: 1-D u must have N_samples entries")
        return _c(u), 0
    if tuple(u.shape) != (N, S):
        raise ValueError(f"sample_pdf: u must be [{N},{S}] or [{S}], got {tuple(u.shape)}")
    if u.stride(0) == 0 and u.stride(1) == 1:
        return u, 0
    u = _c(u)
    return u, S


def sample_pdf_fwd(bins: Tensor, weights: Optional[Tensor], u: Tensor, n_samples: int,
                   bins_are_mids: bool = False, cdf_in: Optional[Tensor] = None,
                   want_inds: bool = False, want_cdf: bool = False, want_std: bool = False):
    bins, bstride = _rows(bins, "sample_pdf: bins")
    N = bins.shape[0]
    M = bins.shape[1] - (1 if bins_are_mids else 0)
    wstride = 0
    if M < 2:
        # (the reference fails here too: its cdf of zero weights is empty - zeros_like(cdf[..., :1]) of an empty cdf,
        # helpers:342-343 - and the gather of helpers:373 raises)
        raise ValueError(f"sample_pdf: {M} bin edge(s): at least two are needed (N_samples >= 3 coarse samples per ray)")
    if weights is not None:
        weights, wstride = _rows(weights, "sample_pdf: weights")
        if tuple(weights.shape) != (N, M - 1):
            raise ValueError(f"sample_pdf: weights must be [{N},{M - 1}], got {tuple(weights.shape)}")
    if cdf_in is not None:
        cdf_in = _c(check(cdf_in, "sample_pdf: cdf"))
    u, ustride = _u_arg(u, N, n_samples)
    dev = bins.device
    samples = torch.empty(N, n_samples, device=dev, dtype=torch.float32)
    inds = torch.empty(N, n_samples, device=dev, dtype=torch.int64) if want_inds else None
    cdf = torch.empty(N, M, device=dev, dtype=torch.float32) if want_cdf else None
    std = torch.empty(N, device=dev, dtype=torch.float32) if want_std else None
    call("scade_sample_pdf_fwd", ptr(bins), bstride, int(bins_are_mids), ptr(weights), wstride,
         ptr(cdf_in), ptr(u), ustride, N, M, n_samples, ptr(samples), ptr(inds), ptr(cdf), ptr(std),
         stream())
    return samples, inds, cdf, std


def sample_pdf_bwd(bins: Tensor, weights: Tensor, u: Tensor, g_samples: Tensor,
                   bins_are_mids: bool = False) -> Tensor:
    bins, bstride = _rows(bins, "sample_pdf.backward: bins")
    N = bins.shape[0]
    M = bins.shape[1] - (1 if bins_are_mids else 0)
    weights, wstride = _rows(weights, "sample_pdf.backward: weights")
    S = g_samples.shape[1]
    u, ustride = _u_arg(u, N, S)
    g_samples = _c(g_samples)
    g_w = torch.empty(N, M - 1, device=bins.device, dtype=torch.float32)
    call("scade_sample_pdf_bwd", ptr(bins), bstride, int(bins_are_mids), ptr(weights), wstride,
         ptr(u), ustride, ptr(g_samples), N, M, S, ptr(g_w), stream())
    return g_w


def merge_sorted(z_a: Tensor, z_b: Tensor, rays: Optional[Tensor] = None):
    z_a, z_b = _c(check(z_a, "merge: z_a")), _c(check(z_b, "merge: z_b"))
    N, Sa = z_a.shape
    Sb = z_b.shape[1]
    out = torch.empty(N, Sa + Sb, device=z_a.device, dtype=torch.float32)
    pts, rstride = None, 0
    if rays is not None:
        rays, rstride = _rows(rays, "merge: rays")
        pts = torch.empty(N, Sa + Sb, 3, device=z_a.device, dtype=torch.float32)
    call("scade_merge_sorted", ptr(z_a), Sa, ptr(z_b), Sb, ptr(rays), rstride, N, ptr(out), ptr(pts),
         stream())
    return out, pts


def ray_tail_supported(S: int, Si: int, merge: bool) -> bool:
    """shapes scade_ray_tail has a kernel for (others go through the three separate entries)"""
    return 3 <= S <= 512 and 1 <= Si <= 1024 and (not merge or (S <= 256 and S + Si <= 512))


def ray_tail(raw: Tensor, z_vals: Tensor, rays: Tensor, noise: Optional[Tensor], u: Tensor, n_samples: int,
             merge: bool, want_pts: bool = True, want_samples: bool = True, want_std: bool = False):
    """scade_ray_tail: raw2outputs -> sample_pdf(z_mid, weights[1:-1], u) [-> sort-merge -> points].
    Returns (rgb, disp, acc, weights, depth, samples|None, z_std|None, z_out|None, pts|None)."""
    check(raw, "ray_tail: raw"); check(z_vals, "ray_tail: z_vals")
    N, S = z_vals.shape
    if tuple(raw.shape) != (N, S, 4):
        raise ValueError(f"ray_tail: raw must be [{N},{S},4], got {tuple(raw.shape)}")
    raw, z_vals = _c(raw), _c(z_vals)
    rays, rstride = _rows(rays, "ray_tail: rays")
    if noise is not None:
        noise = _c(check(noise, "ray_tail: noise"))
    u, ustride = _u_arg(u, N, n_samples)
    dev = raw.device
    rgb = torch.empty(N, 3, device=dev); disp = torch.empty(N, device=dev)
    acc = torch.empty(N, device=dev); w = torch.empty(N, S, device=dev); depth = torch.empty(N, device=dev)
    samples = torch.empty(N, n_samples, device=dev) if (want_samples or not merge) else None
    std = torch.empty(N, device=dev) if want_std else None
    z_out = torch.empty(N, S + n_samples, device=dev) if merge else None
    pts = torch.empty(N, S + n_samples, 3, device=dev) if (merge and want_pts) else None
    call("scade_ray_tail", ptr(raw), ptr(z_vals), ptr(rays), rstride, ptr(noise), N, S, ptr(u), ustride,
         n_samples, ptr(rgb), ptr(disp), ptr(acc), ptr(w), ptr(depth), ptr(samples), ptr(std), ptr(z_out),
         ptr(pts), stream())
    return rgb, disp, acc, w, depth, samples, std, z_out, pts


class CoarseTailFn(torch.autograd.Function):
    """Coarse stage after the MLP (run_scade_scannet.py:660-714): raw2outputs, the detached importance
    samples, the sorted merge and the fine points, one launch.  Differentiable w.r.t. raw through the
    five compositing outputs exactly like CompositeFn; z_vals / pts carry no gradient (:711 detaches)."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, u, n_samples):
        ctx.set_materialize_grads(False)
        rgb, disp, acc, w, depth, _, _, z_out, pts = ray_tail(raw, z_vals, rays, noise, u, n_samples,
                                                              merge=True, want_samples=False)
        ctx.save_for_backward(raw, z_vals, rays, noise if noise is not None else raw.new_empty(0))
        ctx.has_noise = noise is not None
        ctx.mark_non_differentiable(z_out, pts)
        return rgb, disp, acc, w, depth, z_out, pts

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth, *unused):
        raw, z_vals, rays, noise = ctx.saved_tensors
        if all(g is None for g in (g_rgb, g_disp, g_acc, g_w, g_depth)):
            return None, None, None, None, None, None
        g_raw = composite_bwd(raw, z_vals, rays[:, 3:6], noise if ctx.has_noise else None,
                              g_rgb, g_disp, g_acc, g_w, g_depth)
        return g_raw, None, None, None, None, None


class FineTailFn(torch.autograd.Function):
    """Fine stage after the MLP when a gradient is recorded (run_scade_scannet.py:720-730): raw2outputs and
    the depth-hypothesis sampler ``sample_pdf_return_u(z_mid, weights[1:-1], u)`` as ONE forward launch
    (scade_ray_tail) and ONE backward launch (scade_ray_tail_bwd) instead of two operators forward and
    two operators + the slice glue backward.  Differentiable w.r.t. raw; z_std carries no gradient."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, u, n_samples):
        ctx.set_materialize_grads(False)
        rgb, disp, acc, w, depth, samples, std, _, _ = ray_tail(raw, z_vals, rays, noise, u, n_samples, merge=False,
                                                                want_std=True)
        ctx.save_for_backward(raw, z_vals, rays, noise if noise is not None else raw.new_empty(0), u)
        ctx.has_noise, ctx.n_samples = noise is not None, n_samples
        ctx.mark_non_differentiable(std)
        return rgb, disp, acc, w, depth, samples, std

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth, g_samples, _g_std):
        raw, z_vals, rays, noise, u = ctx.saved_tensors
        gs = [g_rgb, g_disp, g_acc, g_w, g_depth]
        if g_samples is None:
            if all(g is None for g in gs):
                return None, None, None, None, None, None
            g_raw = composite_bwd(raw, z_vals, rays[:, 3:6], noise if ctx.has_noise else None, *gs)
            return g_raw, None, None, None, None, None
        N, S = z_vals.shape
        raw_c, z_c = _c(raw), _c(z_vals)
        rays_c, rstride = _rows(rays, "ray_tail.backward: rays")
        u_c, ustride = _u_arg(u, N, ctx.n_samples)
        gs = [None if g is None else _c(g) for g in gs]
        g_samples = _c(g_samples)
        g_raw = torch.empty(N, S, 4, device=raw.device, dtype=torch.float32)
        call("scade_ray_tail_bwd", ptr(raw_c), ptr(z_c), ptr(rays_c), rstride, ptr(noise if ctx.has_noise else None),
             N, S, ptr(u_c), ustride, ctx.n_samples, ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]),
             ptr(g_samples), ptr(g_raw), stream())
        return g_raw, None, None, None, None, None


def gen_rays(H: int, W: int, intrinsic: Tensor, c2w: Tensor, coords: Optional[Tensor] = None,
             near: float = 0.0, far: float = 1.0, image: Optional[Tensor] = None,
             hyps: Optional[Tensor] = None, corner_px: int = 0, edge_px: int = 0,
             want_rows: bool = True, want_od: bool = False, want_mask: bool = False):
    """scade_gen_rays -> dict(rays, rays_o, rays_d, target_s, target_h, mask) (absent = None)."""
    intrinsic = _c(check(intrinsic, "gen_rays: intrinsic").reshape(-1))
    check(c2w, "gen_rays: c2w")
    if c2w.dim() != 2 or c2w.shape[0] < 3 or c2w.shape[1] < 4 or c2w.stride(1) != 1:
        c2w = c2w.reshape(-1, c2w.shape[-1])[:, :4].contiguous()
    dev = c2w.device
    if coords is not None:
        coords = _c(coords.to(device=dev, dtype=torch.int32))
        N = coords.shape[0]
    else:
        N = H * W
    K = 0
    if image is not None:
        image = _c(check(image, "gen_rays: image"))
    if hyps is not None:
        hyps = _c(check(hyps, "gen_rays: hyps"))
        K = hyps.shape[0]
    mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    out = {"rays": mk(N, 11) if want_rows else None,
           "rays_o": mk(N, 3) if want_od else None, "rays_d": mk(N, 3) if want_od else None,
           "target_s": mk(N, 3) if image is not None else None,
           "target_h": mk(K, N) if hyps is not None else None,
           "mask": mk(N) if want_mask else None}
    call("scade_gen_rays", ptr(coords), N, H, W, ptr(intrinsic), ptr(c2w), c2w.stride(0), float(near),
         float(far), ptr(image), ptr(hyps), K, int(corner_px), int(edge_px), ptr(out["rays"]),
         ptr(out["rays_o"]), ptr(out["rays_d"]), ptr(out["target_s"]), ptr(out["target_h"]),
         ptr(out["mask"]), stream())
    return out


def gather_batch(pix: Tensor, H: int, W: int, intrinsic: Tensor, c2w: Tensor, near: float, far: float,
                 image: Optional[Tensor], hyps: Optional[Tensor], rays: Tensor, target_s: Optional[Tensor],
                 target_h: Optional[Tensor], mask: Optional[Tensor] = None, corner_px: int = 0, edge_px: int = 0,
                 scalar=None, tick=None) -> None:
    """scade_gather_batch: the batch of ONE training iteration (run_scade_scannet.py:772-827, :200-219) gathered
    straight INTO the given buffers - the static inputs of a graph-captured step - in one launch that also stores
    ``scalar = (int64 device tensor, value)`` (the view's index) and advances the ``tick`` optimizer states, like
    ``stage_inputs``.  ``pix`` [N] int64 flat pixel indices (row * W + col) on the device; ``intrinsic`` [4],
    ``c2w`` [>= 3, 4], ``image`` [H, W, 3], ``hyps`` [K, H, W(, 1)] of the step's view.  Nothing is allocated."""
    if pix.dtype != torch.int64 or not pix.is_cuda or not pix.is_contiguous():
        raise ValueError("gather_batch: pix must be a contiguous int64 device tensor")
    N = pix.numel()
    check(intrinsic, "gather_batch: intrinsic")
    check(c2w, "gather_batch: c2w")
    if intrinsic.numel() < 4 or not intrinsic.is_contiguous():
        raise ValueError("gather_batch: intrinsic = contiguous [fx, fy, cx, cy]")
    if c2w.dim() != 2 or c2w.shape[0] < 3 or c2w.shape[1] < 4 or c2w.stride(1) != 1:
        raise ValueError("gather_batch: c2w must be [>= 3, >= 4] with unit column stride")
    K = 0
    outs = [(rays, (N, 11), "rays"), (target_s, (N, 3), "target_s"), (mask, (N,), "mask")]
    if image is not None:
        check(image, "gather_batch: image")
        if tuple(image.shape) != (H, W, 3) or not image.is_contiguous():
            raise ValueError("gather_batch: image must be contiguous [H, W, 3]")
    if (hyps is None) != (target_h is None):
        raise ValueError("gather_batch: hyps and target_h go together")
    if hyps is not None:
        check(hyps, "gather_batch: hyps")
        K = hyps.shape[0]
        if hyps.numel() != K * H * W or not hyps.is_contiguous():
            raise ValueError("gather_batch: hyps must be contiguous [K, H, W] or [K, H, W, 1]")
        if target_h.numel() != K * N:
            raise ValueError("gather_batch: target_h must hold [K, N] floats")
        outs.append((target_h, None, "target_h"))
    for t, shape, what in outs:
        if t is None:
            continue
        check(t, "gather_batch: " + what)
        if not t.is_contiguous() or (shape is not None and tuple(t.shape) != shape):
            raise ValueError(f"gather_batch: {what} must be contiguous {shape}")
    sd, sv = (None, 0)
    if scalar is not None:
        if scalar[0].dtype != torch.int64 or not scalar[0].is_cuda:
            raise ValueError("gather_batch: the scalar destination must be an int64 device tensor")
        sd, sv = scalar[0].data_ptr(), int(scalar[1])
    ticks = None
    if tick is not None and any(t is not None for t in tick):
        for t in tick:
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() >= 16 and t.is_contiguous()):
                raise ValueError("gather_batch: tick entries are the float32[16] device states of FusedAdam")
        t2 = (list(tick) + [None])[:2]
        ticks = ctypes.cast((ctypes.c_void_p * 2)(*[None if t is None else t.data_ptr() for t in t2]), ctypes.c_void_p)
    call("scade_gather_batch", ptr(pix), N, int(H), int(W), ptr(intrinsic), ptr(c2w), c2w.stride(0), float(near),
         float(far), ptr(image), ptr(hyps), K, int(corner_px), int(edge_px), ptr(rays), ptr(target_s), ptr(target_h),
         ptr(mask), sd, sv, ticks, stream())


class ResidentBatchGather:
    """``gather_batch`` for a training set that is RESIDENT on the device, prepared once: ``images`` [V, H, W, 3],
    ``hyps`` [V, K, H, W(, 1)], ``poses`` [V, >= 3, 4], ``intrinsics`` [V, 4] (V training views) and the output
    buffers of a captured step are validated here; ``__call__(pix, offset, view)`` is then pointer arithmetic and
    ONE library call (the loop's host cost per iteration: ~10 us).  ``pix`` int64 device tensor, the batch =
    ``pix[offset : offset + N]``.  ``scalar_dst`` receives ``view``; ``tick_states`` = (state, state | None): the
    FusedAdam device states the launch advances (``tick_second=False`` at call time leaves the second alone: the
    scale / shift optimizer after its freeze point)."""

    def __init__(self, H, W, images, hyps, poses, intrinsics, near, far, rays, target_s, target_h, mask=None,
                 corner_px=0, edge_px=0, scalar_dst=None, tick_states=None, points=None, packs=None):
        V = images.shape[0]
        for t, what in ((images, "images"), (hyps, "hyps"), (poses, "poses"), (intrinsics, "intrinsics")):
            check(t, "ResidentBatchGather: " + what)
            if not t.is_contiguous() or t.shape[0] != V:
                raise ValueError(f"ResidentBatchGather: {what} must be contiguous with one entry per view")
        if tuple(images.shape[1:]) != (H, W, 3) or hyps.numel() != V * hyps.shape[1] * H * W:
            raise ValueError("ResidentBatchGather: images [V,H,W,3] / hyps [V,K,H,W] expected")
        if poses.dim() != 3 or poses.shape[1] < 3 or poses.shape[2] != 4 or intrinsics.shape[1:] != (4,):
            raise ValueError("ResidentBatchGather: poses [V, >= 3, 4] / intrinsics [V, 4] expected")
        self.N, self.K, self.V = rays.shape[0], hyps.shape[1], V
        # one validated call against view 0 (shapes of the outputs, dtypes, devices); nothing is launched for N = 0 ...
        gather_batch(torch.zeros(self.N, dtype=torch.int64, device=rays.device), H, W, intrinsics[0], poses[0], near,
                     far, images[0], hyps[0], rays, target_s, target_h, mask, corner_px, edge_px)
        if scalar_dst is not None and (scalar_dst.dtype != torch.int64 or not scalar_dst.is_cuda):
            raise ValueError("ResidentBatchGather: the scalar destination must be an int64 device tensor")
        self._keep = (images, hyps, poses, intrinsics, rays, target_s, target_h, mask, scalar_dst, tick_states)
        self._img = (images.data_ptr(), H * W * 3 * 4)
        self._hyp = (hyps.data_ptr(), self.K * H * W * 4)
        self._pose = (poses.data_ptr(), poses.shape[1] * 4 * 4)
        self._intr = (intrinsics.data_ptr(), 16)
        self._fixed_a = (self.N, int(H), int(W))
        self._fixed_b = (4, float(near), float(far))
        self._fixed_c = (self.K, int(corner_px), int(edge_px), ptr(rays), ptr(target_s), ptr(target_h), ptr(mask),
                         None if scalar_dst is None else scalar_dst.data_ptr())
        self._ticks = {True: None, False: None}
        self._pix_ok, self._tick_src, self._tick_ptrs = None, None, None
        if tick_states is not None and any(t is not None for t in tick_states):
            # entries: a FusedAdam (its ``.state`` is looked up at every call: use_device_state() / a reloaded state
            # REPLACES that tensor) or the float32[16] state tensor itself
            self._tick_src = (list(tick_states) + [None])[:2]
            self._bind_ticks()
        # ``points`` (a CoarsePoints of this batch size): the captured step's coarse samples are computed in this launch
        if points is not None and points.N != self.N:
            raise ValueError("ResidentBatchGather: the coarse-sample buffers are for another batch size")
        if packs is not None and points is None:
            raise ValueError("ResidentBatchGather: packs ride with the coarse points (points=)")
        self._points, self._packs = points, packs
        self._fn = getattr(load(), "scade_gather_batch_points" if points is not None else "scade_gather_batch")

    def _tick_tensors(self):
        return [t if (t is None or torch.is_tensor(t)) else t.state for t in self._tick_src]

    def _bind_ticks(self):
        t2 = self._tick_tensors()
        for t in t2:
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() >= 16 and t.is_contiguous()):
                raise ValueError("ResidentBatchGather: tick_states are the float32[16] device states of FusedAdam")
        p2 = [None if t is None else t.data_ptr() for t in t2]
        self._tick_ptrs = tuple(p2)
        self._tick_arrs = ((ctypes.c_void_p * 2)(*p2), (ctypes.c_void_p * 2)(p2[0], None))
        self._ticks = {True: ctypes.cast(self._tick_arrs[0], ctypes.c_void_p),
                       False: ctypes.cast(self._tick_arrs[1], ctypes.c_void_p)}

    def __call__(self, pix: Tensor, offset: int, view: int, tick_second: bool = True) -> None:
        if not 0 <= view < self.V:
            raise IndexError(f"ResidentBatchGather: view {view} outside [0, {self.V})")
        if pix.dtype != torch.int64 or offset < 0 or offset + self.N > pix.numel():
            raise ValueError("ResidentBatchGather: pix[offset : offset + N] must lie inside an int64 tensor")
        key = (pix.data_ptr(), pix.numel())
        if key != self._pix_ok:
            # (once per new buffer - the loop hands over the same permutation for H * W / N steps: a host tensor or a
            # strided view would reach the kernel as a wild device pointer)
            if not pix.is_cuda or pix.device != self._keep[4].device or not pix.is_contiguous():
                raise ValueError("ResidentBatchGather: pix must be a contiguous int64 tensor on the batch's device")
            self._pix_ok = key
        if self._tick_ptrs is not None and \
                tuple(None if t is None else t.data_ptr() for t in self._tick_tensors()) != self._tick_ptrs:
            self._bind_ticks()            # an optimizer's device state was replaced since the last call
        rc = self._fn(pix.data_ptr() + 8 * offset, *self._fixed_a, self._intr[0] + view * self._intr[1],
                      self._pose[0] + view * self._pose[1], *self._fixed_b, self._img[0] + view * self._img[1],
                      self._hyp[0] + view * self._hyp[1], *self._fixed_c, view, self._ticks[bool(tick_second)],
                      *(() if self._points is None else
                        self._points.tail_args() + (_NO_PACK if self._packs is None else self._packs.tail_args())), stream())
        if rc != 0:
            raise RuntimeError(f"scade_gather_batch failed (code {rc}): {last_error()}")


# ---------------------------------------------------------------------------
# autograd glue
# ---------------------------------------------------------------------------

class CompositeFn(torch.autograd.Function):
    """raw2outputs (run_scade_scannet.py:530-562); differentiable w.r.t. raw."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise):
        ctx.set_materialize_grads(False)     # unused outputs arrive as None, not as zero-filled tensors
        outs = composite_fwd(raw, z_vals, rays_d, noise)
        ctx.save_for_backward(raw, z_vals, rays_d, noise if noise is not None else raw.new_empty(0))
        ctx.has_noise = noise is not None
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z_vals, rays_d, noise = ctx.saved_tensors
        if all(g is None for g in (g_rgb, g_disp, g_acc, g_w, g_depth)):
            return None, None, None, None
        g_raw = composite_bwd(raw, z_vals, rays_d, noise if ctx.has_noise else None,
                              g_rgb, g_disp, g_acc, g_w, g_depth)
        return g_raw, None, None, None


class SamplePdfFn(torch.autograd.Function):
    """sample_pdf / sample_pdf_return_u (helpers:337-436); differentiable w.r.t. weights."""

    @staticmethod
    def forward(ctx, bins, weights, u, bins_are_mids, want_std):
        ctx.set_materialize_grads(False)
        samples, _, _, std = sample_pdf_fwd(bins, weights, u, u.shape[-1], bins_are_mids,
                                            want_std=want_std)
        ctx.save_for_backward(bins, weights, u)
        ctx.mids = bins_are_mids
        if want_std:
            ctx.mark_non_differentiable(std)
            return samples, std
        return samples

    @staticmethod
    def backward(ctx, g_samples, *unused):
        bins, weights, u = ctx.saved_tensors
        if g_samples is None:
            return None, None, None, None, None
        g_w = sample_pdf_bwd(bins, weights, u, g_samples, ctx.mids)
        return None, g_w, None, None, None


def _carve_per_sample(hyp: Tensor, N: int, P: int) -> bool:
    """target_hypothesis [K,N,1] (one hypothesis per ray, repeated over the samples, helpers:97-99) or
    [K,N,P] (cached quantiles: a hypothesis per sample, helpers:100-102)."""
    if hyp.dim() != 3 or hyp.shape[1] != N or hyp.shape[2] not in (1, P):
        raise ValueError(f"space_carving: target_hypothesis must be [K,{N},1] or [K,{N},{P}], got {tuple(hyp.shape)}")
    return hyp.shape[2] != 1


class CarveFn(torch.autograd.Function):
    """compute_space_carving_loss (helpers:93-128)."""

    @staticmethod
    def forward(ctx, pred, hyp, mask, threshold, is_joint):
        check(pred, "space_carving: pred_depth"); check(hyp, "space_carving: target_hypothesis")
        N, P = pred.shape
        K = hyp.shape[0]
        knp = _carve_per_sample(hyp, N, P)          # hypotheses cached per sample [K,N,P] (helpers:100-102)
        pred_c, hyp_c = _c(pred), _c(hyp.reshape(K, N, P) if knp else hyp.reshape(K, N))
        mask_c = None if mask is None else _c(check(mask, "space_carving: mask").reshape(N))
        nws = int(_lib.load().scade_carve_workspace_floats(N, P, K, int(is_joint)))
        ws = torch.empty(nws, device=pred.device, dtype=torch.float32)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        call("scade_carve_knp_fwd" if knp else "scade_carve_fwd", ptr(pred_c), ptr(hyp_c), ptr(mask_c),
             float(threshold), int(is_joint), N, P, K, ptr(ws), ptr(loss), stream())
        ctx.save_for_backward(pred_c, hyp_c, mask_c if mask_c is not None else pred.new_empty(0), ws)
        ctx.cfg = (float(threshold), int(is_joint), mask is not None, tuple(hyp.shape), knp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, hyp, mask, ws = ctx.saved_tensors
        thr, joint, has_mask, hyp_shape, knp = ctx.cfg
        N, P = pred.shape
        K = hyp.shape[0]
        g = _c(g.reshape(1).to(torch.float32))
        g_pred = torch.empty_like(pred)
        g_hyp = torch.empty_like(hyp)
        call("scade_carve_knp_bwd" if knp else "scade_carve_bwd", ptr(pred), ptr(hyp),
             ptr(mask if has_mask else None), thr, joint, N, P, K, ptr(ws), ptr(g), ptr(g_pred), ptr(g_hyp), stream())
        return g_pred, g_hyp.reshape(hyp_shape), None, None, None


class CarveJointShardedFn(torch.autograd.Function):
    """is_joint=True space-carving loss of a RAY-SHARDED batch: the mean over
    rays runs over all shards BEFORE the min over the K hypotheses, so the [K,P] column means are
    combined across ranks (parallel.combine_shard_means: one all-reduce of K*P floats) between the two
    kernel phases.  Every rank returns the GLOBAL loss; the backward yields this shard's part of its
    gradient, so the SUM over ranks (the trainer's gradient all-reduce) is the gradient of the loss."""

    @staticmethod
    def forward(ctx, pred, hyp, mask, threshold, group, n_total=None):
        from . import parallel
        check(pred, "space_carving: pred_depth"); check(hyp, "space_carving: target_hypothesis")
        N, P = pred.shape
        K = hyp.shape[0]
        knp = _carve_per_sample(hyp, N, P)
        pred_c, hyp_c = _c(pred), _c(hyp.reshape(K, N, P) if knp else hyp.reshape(K, N))
        mask_c = None if mask is None else _c(check(mask, "space_carving: mask").reshape(N))
        ws = torch.empty(int(_lib.load().scade_carve_workspace_floats(N, P, K, 1)), device=pred.device,
                         dtype=torch.float32)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        call("scade_carve_knp_joint_colmean" if knp else "scade_carve_joint_colmean", ptr(pred_c), ptr(hyp_c),
             ptr(mask_c), float(threshold), N, P, K, ptr(ws), stream())
        share, _ = parallel.combine_shard_means(ws[:K * P], N, group, n_total)
        call("scade_carve_joint_min", ptr(ws), P, K, ptr(loss), stream())
        ctx.save_for_backward(pred_c, hyp_c, mask_c if mask_c is not None else pred.new_empty(0), ws)
        # the kernel's backward divides by this shard's N; the global mean divides by N_total
        ctx.cfg = (float(threshold), mask is not None, tuple(hyp.shape), share, knp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, hyp, mask, ws = ctx.saved_tensors
        thr, has_mask, hyp_shape, factor, knp = ctx.cfg
        N, P = pred.shape
        K = hyp.shape[0]
        g = _c((g.reshape(1) * factor).to(torch.float32))
        g_pred = torch.empty_like(pred)
        g_hyp = torch.empty_like(hyp)
        call("scade_carve_knp_bwd" if knp else "scade_carve_bwd", ptr(pred), ptr(hyp),
             ptr(mask if has_mask else None), thr, 1, N, P, K, ptr(ws), ptr(g), ptr(g_pred), ptr(g_hyp), stream())
        return g_pred, g_hyp.reshape(hyp_shape), None, None, None, None


class TrainLossFn(torch.autograd.Function):
    """loss = mse(rgb, target) + w * space_carving(pred, hyp * scale[img] + shift[img]) + mse(rgb0, target)
    (run_scade_scannet.py:954, :968-983) as ONE forward and ONE backward entry (scade_train_loss_*): what the
    Trainer uses instead of ~25 launches of separate operators.  Returns (total, [img_loss, carve,
    img_loss0]).  The scale / shift gradients are accumulated straight into ``scales.grad`` /
    ``shifts.grad`` when those exist (the Trainer's gradient bucket), otherwise returned densely."""

    @staticmethod
    def forward(ctx, rgb, rgb0, target, pred, hyp, scales, shifts, img_i, mask, mse_masked, carve_on,
                carve_weight, threshold, out_scale):
        for t, w in ((rgb, "rgb"), (rgb0, "rgb0"), (target, "target"), (pred, "pred_hyp"), (hyp, "hypotheses")):
            check(t, "train_loss: " + w)
        N, P = pred.shape
        K = hyp.shape[0]
        if tuple(hyp.shape[1:]) not in ((N,), (N, 1)):
            raise ValueError(f"train_loss: hypotheses must be [K,{N},1], got {tuple(hyp.shape)}")
        if tuple(rgb.shape) != (N, 3) or tuple(rgb0.shape) != (N, 3) or tuple(target.shape) != (N, 3):
            raise ValueError("train_loss: rgb, rgb0 and target must be [N,3]")
        rgb_c, rgb0_c, tgt_c, pred_c, hyp_c = _c(rgb), _c(rgb0), _c(target), _c(pred), _c(hyp.reshape(K, N))
        sc, sh = _c(scales.detach().reshape(-1)), _c(shifts.detach().reshape(-1))
        mask_c = None if mask is None else _c(check(mask, "train_loss: mask").reshape(N))
        idx_t = None
        if torch.is_tensor(img_i):
            # the kernel reads it as one int64 on the device and indexes scales / shifts / their gradients with it
            if img_i.dtype != torch.int64 or img_i.device != pred.device or img_i.numel() != 1:
                raise TypeError(f"train_loss: a tensor img_i must be ONE int64 on {pred.device}, got "
                                f"{img_i.dtype} x {img_i.numel()} on {img_i.device}")
            idx_t = _c(img_i.reshape(1))
        # beside a device index the ABI's img_i argument is its bound (n_images)
        idx = sc.numel() if idx_t is not None else int(img_i)
        if idx_t is None and not 0 <= idx < sc.numel():
            raise IndexError(f"train_loss: img_i {idx} outside [0, {sc.numel()})")
        ws = torch.empty(4 * N, device=pred.device, dtype=torch.float32)
        loss4 = torch.empty(4, device=pred.device, dtype=torch.float32)
        call("scade_train_loss_fwd", ptr(rgb_c), ptr(rgb0_c), ptr(tgt_c), ptr(pred_c), ptr(hyp_c), ptr(sc), ptr(sh),
             ptr(idx_t), idx, ptr(mask_c), int(bool(mse_masked)), int(bool(carve_on)), float(carve_weight),
             float(threshold), float(out_scale), N, P, K, ptr(ws), ptr(loss4), stream())
        ctx.save_for_backward(rgb_c, rgb0_c, tgt_c, pred_c, hyp_c, sc, sh,
                              mask_c if mask_c is not None else pred.new_empty(0),
                              idx_t if idx_t is not None else pred.new_empty(0, dtype=torch.long), ws)
        ctx.cfg = (idx, mask is not None, idx_t is not None, int(bool(mse_masked)), int(bool(carve_on)),
                   float(carve_weight), float(threshold), float(out_scale), tuple(hyp.shape))
        ctx.ss = (scales, shifts)
        comps = loss4[1:]
        ctx.mark_non_differentiable(comps)
        ctx.set_materialize_grads(False)      # no zero-fill launch for the non-differentiable second output
        return loss4[0], comps

    @staticmethod
    def backward(ctx, g, _unused):
        rgb, rgb0, tgt, pred, hyp, sc, sh, mask, idx_t, ws = ctx.saved_tensors
        idx, has_mask, has_idx, mse_masked, carve_on, w, thr, out_scale, hyp_shape = ctx.cfg
        scales, shifts = ctx.ss
        N, P = pred.shape
        K = hyp.shape[0]
        g = _c(g.reshape(1).to(torch.float32))
        g_rgb, g_rgb0, g_pred = torch.empty_like(rgb), torch.empty_like(rgb0), torch.empty_like(pred)

        def sink(p):     # accumulate straight into an existing contiguous fp32 .grad (the Trainer's bucket view)
            gr = p.grad
            return gr if (gr is not None and gr.is_contiguous() and gr.dtype == torch.float32
                          and gr.numel() == p.numel()) else None
        gs, gh = sink(scales), sink(shifts)
        direct = gs is not None and gh is not None
        if not direct:
            gs, gh = torch.zeros_like(sc), torch.zeros_like(sh)
        call("scade_train_loss_bwd", ptr(rgb), ptr(rgb0), ptr(tgt), ptr(pred), ptr(hyp), ptr(sc), ptr(sh),
             ptr(idx_t if has_idx else None), idx, ptr(mask if has_mask else None), mse_masked, carve_on, w, thr,
             out_scale, N, P, K, ptr(ws), ptr(g), ptr(g_rgb), ptr(g_rgb0), ptr(g_pred), ptr(gs), ptr(gh), stream())
        need = ctx.needs_input_grad
        return (g_rgb if need[0] else None, g_rgb0 if need[1] else None, None, g_pred if need[3] else None, None,
                None if direct or not need[5] else gs.reshape(scales.shape),
                None if direct or not need[6] else gh.reshape(shifts.shape),
                None, None, None, None, None, None, None)


class TrainLossUnitFn(torch.autograd.Function):
    """TrainLossFn with forward AND backward in ONE pair of launches (scade_train_loss_fb) for the train step,
    which differentiates the total with a unit gradient: the gradients w.r.t. rgb / rgb0 / pred are computed while
    the loss terms are, the scale / shift rows of the gradient bucket are WRITTEN by the reduce (all n_images rows:
    no zero fill of them needed), and backward() only hands the saved tensors out.  ``unit`` is the tensor the
    caller passes to ``backward(gradient=unit)``: any other incoming gradient raises (it would need the scale /
    shift rows re-done).  Same arithmetic as TrainLossFn."""

    @staticmethod
    def forward(ctx, rgb, rgb0, target, pred, hyp, scales, shifts, img_i, mask, mse_masked, carve_on,
                carve_weight, threshold, out_scale, unit):
        for t, w in ((rgb, "rgb"), (rgb0, "rgb0"), (target, "target"), (pred, "pred_hyp"), (hyp, "hypotheses")):
            check(t, "train_loss: " + w)
        N, P = pred.shape
        K = hyp.shape[0]
        if tuple(hyp.shape[1:]) not in ((N,), (N, 1)):
            raise ValueError(f"train_loss: hypotheses must be [K,{N},1], got {tuple(hyp.shape)}")
        if tuple(rgb.shape) != (N, 3) or tuple(rgb0.shape) != (N, 3) or tuple(target.shape) != (N, 3):
            raise ValueError("train_loss: rgb, rgb0 and target must be [N,3]")
        gs, gh = scales.grad, shifts.grad
        ok = lambda g, p: g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.numel() == p.numel()
        if not (ok(gs, scales) and ok(gh, shifts)):
            raise RuntimeError("train_loss (unit-gradient form): scales / shifts need contiguous fp32 .grad buffers "
                               "(the Trainer's gradient bucket)")
        rgb_c, rgb0_c, tgt_c, pred_c, hyp_c = _c(rgb), _c(rgb0), _c(target), _c(pred), _c(hyp.reshape(K, N))
        sc, sh = _c(scales.detach().reshape(-1)), _c(shifts.detach().reshape(-1))
        mask_c = None if mask is None else _c(check(mask, "train_loss: mask").reshape(N))
        idx_t = None
        if torch.is_tensor(img_i):
            if img_i.dtype != torch.int64 or img_i.device != pred.device or img_i.numel() != 1:
                raise TypeError(f"train_loss: a tensor img_i must be ONE int64 on {pred.device}, got "
                                f"{img_i.dtype} x {img_i.numel()} on {img_i.device}")
            idx_t = _c(img_i.reshape(1))
        idx = sc.numel() if idx_t is not None else int(img_i)
        if idx_t is None and not 0 <= idx < sc.numel():
            raise IndexError(f"train_loss: img_i {idx} outside [0, {sc.numel()})")
        dev = pred.device
        ws = torch.empty(8 * N, device=dev, dtype=torch.float32)
        loss4 = torch.empty(4, device=dev, dtype=torch.float32)
        g_rgb, g_rgb0, g_pred = torch.empty_like(rgb_c), torch.empty_like(rgb0_c), torch.empty_like(pred_c)
        call("scade_train_loss_fb", ptr(rgb_c), ptr(rgb0_c), ptr(tgt_c), ptr(pred_c), ptr(hyp_c), ptr(sc), ptr(sh),
             ptr(idx_t), idx, ptr(mask_c), int(bool(mse_masked)), int(bool(carve_on)), float(carve_weight),
             float(threshold), float(out_scale), N, P, K, ptr(ws), ptr(loss4), ptr(g_rgb), ptr(g_rgb0), ptr(g_pred),
             ptr(gs), ptr(gh), sc.numel(), stream())
        ctx.save_for_backward(g_rgb, g_rgb0, g_pred)
        ctx.unit_ptr, ctx.unit_version = unit.data_ptr(), unit._version
        comps = loss4[1:]
        ctx.mark_non_differentiable(comps)
        ctx.set_materialize_grads(False)
        return loss4[0], comps

    @staticmethod
    def backward(ctx, g, _unused):
        if g is None or g.data_ptr() != ctx.unit_ptr or g.numel() != 1 or g._version != ctx.unit_version:
            raise RuntimeError("train_loss (unit-gradient form): backward() must be called with the unit tensor "
                               "given to the forward (Trainer.backward does); use ops.TrainLossFn otherwise")
        g_rgb, g_rgb0, g_pred = ctx.saved_tensors
        need = ctx.needs_input_grad
        return (g_rgb if need[0] else None, g_rgb0 if need[1] else None, None, g_pred if need[3] else None) + (None,) * 11


class FineTailLossFn(torch.autograd.Function):
    """The fine tail (FineTailFn), the unit-gradient train loss (TrainLossUnitFn) and the backward of BOTH tails of a
    train step as ONE launch + the loss's reduce (scade_ray_tail_train): everything - the loss terms, the scale /
    shift rows of the gradient bucket, d loss / d raw of the fine AND the coarse network - is computed in forward();
    backward() hands the two gradients to the MLP backward.  Same device functions, same bits as the separate
    operators.  ``rgb0`` is the coarse colour as a VALUE (pass it detached: its gradient is applied here, through
    ``raw0``); ``unit`` as in TrainLossUnitFn."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays, u, n_samples, raw0, z0, rgb0, target, hyp, scales, shifts, img_i, mask,
                mse_masked, carve_on, carve_weight, threshold, out_scale, unit, want_gmax=False):
        for t, w in ((raw, "raw"), (z_vals, "z_vals"), (raw0, "raw0"), (z0, "z_vals0"), (rgb0, "rgb0"),
                     (target, "target"), (hyp, "hypotheses")):
            check(t, "ray_tail_train: " + w)
        N, S = z_vals.shape
        S0 = z0.shape[1]
        K = hyp.shape[0]
        if tuple(raw.shape) != (N, S, 4) or tuple(raw0.shape) != (N, S0, 4) or tuple(z0.shape) != (N, S0):
            raise ValueError("ray_tail_train: raw [N,S,4], raw0 [N,S0,4], z_vals0 [N,S0] expected")
        if tuple(hyp.shape[1:]) not in ((N,), (N, 1)):
            raise ValueError(f"ray_tail_train: hypotheses must be [K,{N},1], got {tuple(hyp.shape)}")
        if tuple(rgb0.shape) != (N, 3) or tuple(target.shape) != (N, 3):
            raise ValueError("ray_tail_train: rgb0 and target must be [N,3]")
        gs, gh = scales.grad, shifts.grad
        ok = lambda g, p: g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.numel() == p.numel()
        if not (ok(gs, scales) and ok(gh, shifts)):
            raise RuntimeError("ray_tail_train: scales / shifts need contiguous fp32 .grad buffers (the Trainer's "
                               "gradient bucket)")
        raw_c, z_c, raw0_c, z0_c = _c(raw.detach()), _c(z_vals), _c(raw0.detach()), _c(z0)
        rays_c, rstride = _rows(rays, "ray_tail_train: rays")
        u_c, ustride = _u_arg(u, N, n_samples)
        rgb0_c, tgt_c, hyp_c = _c(rgb0), _c(target), _c(hyp.reshape(K, N))
        sc, sh = _c(scales.detach().reshape(-1)), _c(shifts.detach().reshape(-1))
        mask_c = None if mask is None else _c(check(mask, "ray_tail_train: mask").reshape(N))
        idx_t = None
        if torch.is_tensor(img_i):
            if img_i.dtype != torch.int64 or img_i.device != raw.device or img_i.numel() != 1:
                raise TypeError(f"ray_tail_train: a tensor img_i must be ONE int64 on {raw.device}, got "
                                f"{img_i.dtype} x {img_i.numel()} on {img_i.device}")
            idx_t = _c(img_i.reshape(1))
        idx = sc.numel() if idx_t is not None else int(img_i)
        if idx_t is None and not 0 <= idx < sc.numel():
            raise IndexError(f"ray_tail_train: img_i {idx} outside [0, {sc.numel()})")
        dev = raw.device
        f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        rgb, disp, acc, w, depth = f(N, 3), f(N), f(N), f(N, S), f(N)
        samples, std = f(N, n_samples), f(N)
        ws, loss4 = f(8 * N), f(4)
        g_raw, g_raw0 = f(N, S, 4), f(N, S0, 4)
        t0 = KERNEL_TIMER.start() if KERNEL_TIMER is not None else None
        GMAX_READY.clear()
        if want_gmax:
            # the loss-scale maxima of g_raw / g_raw0 come out of this launch's reduce: the joint 16-bit backward of
            # the step (mlp_bwd.flush_deferred) finds them by the gradients' addresses and skips its own maxima launch
            gws, gm = f(2 * N), f(2, 256)
            call("scade_ray_tail_train_gmax", ptr(raw_c), ptr(z_c), ptr(rays_c), rstride, N, S, ptr(u_c), ustride,
                 n_samples, ptr(rgb), ptr(disp), ptr(acc), ptr(w), ptr(depth), ptr(samples), ptr(std),
                 ptr(rgb0_c), ptr(tgt_c), ptr(hyp_c), ptr(sc), ptr(sh), ptr(idx_t), idx, ptr(mask_c),
                 int(bool(mse_masked)), int(bool(carve_on)), float(carve_weight), float(threshold), float(out_scale),
                 K, ptr(ws), ptr(loss4), ptr(gs), ptr(gh), sc.numel(), ptr(g_raw), ptr(raw0_c), ptr(z0_c), S0,
                 ptr(g_raw0), ptr(gws), ptr(gm[0]), ptr(gm[1]), stream())
            GMAX_READY[g_raw.data_ptr()] = gm[0]
            GMAX_READY[g_raw0.data_ptr()] = gm[1]
        else:
            call("scade_ray_tail_train", ptr(raw_c), ptr(z_c), ptr(rays_c), rstride, None, N, S, ptr(u_c), ustride,
                 n_samples, ptr(rgb), ptr(disp), ptr(acc), ptr(w), ptr(depth), ptr(samples), ptr(std),
                 ptr(rgb0_c), ptr(tgt_c), ptr(hyp_c), ptr(sc), ptr(sh), ptr(idx_t), idx, ptr(mask_c),
                 int(bool(mse_masked)), int(bool(carve_on)), float(carve_weight), float(threshold), float(out_scale),
                 K, ptr(ws), ptr(loss4), ptr(gs), ptr(gh), sc.numel(), ptr(g_raw), ptr(raw0_c), ptr(z0_c), None, S0,
                 ptr(g_raw0), stream())
        if t0 is not None:
            KERNEL_TIMER.stop("ray_tail_train", t0, 0.0)
        ctx.save_for_backward(g_raw, g_raw0)
        ctx.unit_ptr, ctx.unit_version = unit.data_ptr(), unit._version
        comps = loss4[1:]
        ctx.mark_non_differentiable(comps, rgb, disp, acc, w, depth, samples, std)
        ctx.set_materialize_grads(False)
        return loss4[0], comps, rgb, disp, acc, w, depth, samples, std

    @staticmethod
    def backward(ctx, g, *_unused):
        if g is None or g.data_ptr() != ctx.unit_ptr or g.numel() != 1 or g._version != ctx.unit_version:
            raise RuntimeError("ray_tail_train: backward() must be called with the unit tensor given to the forward "
                               "(Trainer.backward does)")
        g_raw, g_raw0 = ctx.saved_tensors
        need = ctx.needs_input_grad
        return (g_raw if need[0] else None, None, None, None, None, g_raw0 if need[5] else None) + (None,) * 15


class MseFn(torch.autograd.Function):
    """img2mse (helpers:11), optional per-row mask (run_scade_wild.py:978-986)."""

    @staticmethod
    def forward(ctx, x, y, mask):
        check(x, "img2mse: x"); check(y, "img2mse: y")
        if x.shape != y.shape:
            raise ValueError(f"img2mse: shapes differ: {tuple(x.shape)} vs {tuple(y.shape)}")
        xc, yc = _c(x), _c(y)
        c = x.shape[-1] if x.dim() > 1 else 1
        n = x.numel() // c
        mc = None if mask is None else _c(check(mask, "img2mse: mask").reshape(n))
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        call("scade_mse_fwd", ptr(xc), ptr(yc), ptr(mc), n, c, ptr(loss), stream())
        ctx.save_for_backward(xc, yc, mc if mc is not None else x.new_empty(0))
        ctx.dims = (n, c, mask is not None, x.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, y, mask = ctx.saved_tensors
        n, c, has_mask, shape = ctx.dims
        g = _c(g.reshape(1).to(torch.float32))
        gx = torch.empty_like(x)
        call("scade_mse_bwd", ptr(x), ptr(y), ptr(mask if has_mask else None), n, c, ptr(g), ptr(gx),
             stream())
        gx = gx.reshape(shape)
        return (gx if ctx.needs_input_grad[0] else None,
                (-gx) if ctx.needs_input_grad[1] else None, None)
