"""Fused Adam over FlatParams (scade_adam_step): the optimizer.step() of the reference's
train loop (run_scade_scannet.py:469, :888, :993-997) as one kernel launch."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import call, check, ptr, stream
from .parallel import FlatParams


class FusedAdam:
    def __init__(self, flat: FlatParams, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
        check(flat.data, "FusedAdam: params")
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.steps = 0

    def step(self, grad_scale: float = 1.0, lr=None):
        self.steps += 1
        call("scade_adam_step", ptr(self.flat.data), ptr(self.flat.grad), ptr(self.exp_avg),
             ptr(self.exp_avg_sq), self.flat.numel, float(self.lr if lr is None else lr),
             float(self.betas[0]), float(self.betas[1]), float(self.eps), self.steps,
             float(grad_scale), stream())
        # parameters were updated through a raw pointer (tensor version counters did not
        # move): advance the global epoch so NeRF.packed()/packed_t() re-pack
        ops.PARAM_EPOCH += 1

    def use_device_state(self, decay_rate: float = 1.0, decay_step: int = 0, grad_scale: float = 1.0,
                         iter_offset: int = 0):
        """Move step count, staircase learning rate and bias corrections to the device
        (``step_dev``): no per-step launch argument is left, so a captured graph stays valid.
        The staircase runs on the reference's loop index i = steps taken so far + 1 + ``iter_offset``
        (run_scade_scannet.py:899-900, :988)."""
        self.state = torch.zeros(16, device=self.flat.data.device, dtype=torch.float32)
        self.state[:8] = torch.tensor([float(self.steps), self.lr, decay_rate, float(decay_step), self.betas[0],
                                       self.betas[1], self.eps, grad_scale], dtype=torch.float32)
        self.state[11] = float(iter_offset)
        return self

    def step_dev(self):
        call("scade_adam_step_dev", ptr(self.flat.data), ptr(self.flat.grad), ptr(self.exp_avg),
             ptr(self.exp_avg_sq), self.flat.numel, ptr(self.state), stream())
        self.steps += 1
        ops.PARAM_EPOCH += 1

    def zero_grad(self):
        self.flat.zero_grad()


def adam_step_pair(a: FusedAdam, b: Optional[FusedAdam], lr_a=None, dev: bool = False, ticked: bool = False):
    """``a.step(lr=lr_a)`` and (``b`` given) ``b.step()`` as ONE update launch (scade_adam_step2); ``dev``: both
    optimizers read their scalars from their device state (``use_device_state``; graph-captured steps);
    ``ticked``: that state was already advanced for this step (``ops.stage_inputs(tick=...)``)."""
    import ctypes
    opts = [a] + ([b] if b is not None else [])
    for o in opts:
        o.steps += 1
    two = lambda f, ct: (ct * 2)(*[f(o) for o in opts] + ([ct()] if len(opts) == 1 else []))
    P = ctypes.c_void_p
    pp = lambda f: ctypes.cast(two(lambda o: f(o).data_ptr(), ctypes.c_void_p), P)
    n = two(lambda o: o.flat.numel, ctypes.c_long)
    if dev:
        st = pp(lambda o: o.state)
        call("scade_adam_step2", pp(lambda o: o.flat.data), pp(lambda o: o.flat.grad), pp(lambda o: o.exp_avg),
             pp(lambda o: o.exp_avg_sq), ctypes.cast(n, P), None, None, None, None, None, None, st, int(bool(ticked)), stream())
    else:
        lrs = two(lambda o: float(lr_a if (o is a and lr_a is not None) else o.lr), ctypes.c_float)
        b1 = two(lambda o: float(o.betas[0]), ctypes.c_float)
        b2 = two(lambda o: float(o.betas[1]), ctypes.c_float)
        eps = two(lambda o: float(o.eps), ctypes.c_float)
        stp = two(lambda o: int(o.steps), ctypes.c_int)
        gs = two(lambda o: 1.0, ctypes.c_float)
        call("scade_adam_step2", pp(lambda o: o.flat.data), pp(lambda o: o.flat.grad), pp(lambda o: o.exp_avg),
             pp(lambda o: o.exp_avg_sq), ctypes.cast(n, P), ctypes.cast(lrs, P), ctypes.cast(b1, P),
             ctypes.cast(b2, P), ctypes.cast(eps, P), ctypes.cast(stp, P), ctypes.cast(gs, P), None, 0, stream())
    ops.PARAM_EPOCH += 1
