"""Test-only CPU model of the 16-bit ("lp") MLP kernels: the oracle's NeRF.forward with the
kernel's rounding points (operands of every MFMA layer rounded to fp16 / bf16, fp32 accumulate,
fp32 biases and heads) and straight-through gradients.  A ReLU network's gradient depends on
its sign pattern, so the mixed-precision backward is compared with autograd through THIS
function (same sign pattern), not through the fp32 oracle."""
import torch
import torch.nn.functional as F

from oracle import scade_oracle as O


def _q(t, dtype):
    """round to dtype, identity gradient"""
    return t + (t.to(dtype).to(t.dtype) - t).detach()


def nerf_forward_lp(p, x, dtype, input_ch=57):
    q = lambda t: _q(t, dtype)
    pts, views = q(x[..., :input_ch]), q(x[..., input_ch:])
    h = pts
    for i in range(O.D_LAYERS):
        h = q(F.relu(F.linear(h, q(p[f"pts_linears.{i}.weight"]), p[f"pts_linears.{i}.bias"])))
        if i == O.SKIP:
            h = torch.cat([pts, h], -1)
    alpha = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
    feat = q(F.linear(h, q(p["feature_linear.weight"]), p["feature_linear.bias"]))
    h = torch.cat([feat, views], -1)
    h = q(F.relu(F.linear(h, q(p["views_linears.0.weight"]), p["views_linears.0.bias"])))
    rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])
    return torch.cat([rgb, F.softplus(alpha, beta=10)], -1)
