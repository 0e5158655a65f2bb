#!/usr/bin/env python3
"""Gradient accuracy + speed of the 16-bit (f16 / bf16) training kernels vs the exact fp32 kernels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from scade_amd import ops
from oracle import scade_oracle as O

dev = torch.device("cuda:0")
params = O.nerf_init(5)
def make():
    net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
    net.load_state_dict(params); return net.to(dev)
torch.manual_seed(9)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
pts = torch.rand(P, 3) * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)
x = torch.cat([O.embed(pts, 9), vd], -1).to(dev)
G = (torch.randn(P, 4) * torch.logspace(-6, 0, P)[:, None] * 1e-3).to(dev)
G[::7] = 0.0
res = {}
for prec in ("f32", "f16", "bf16"):
    net = make(); net.train_precision = prec
    out = net(x)
    (out * G).sum().backward()
    torch.cuda.synchronize()
    res[prec] = ({k: p.grad.clone() for k, p in net.named_parameters()}, out.detach())
ref, oref = res["f32"]
for prec in ("f16", "bf16"):
    got, o = res[prec]
    print(f"== {prec}: forward rel-L2 {float((o - oref).norm() / oref.norm()):.3e}")
    worst = 0.0
    for k in ref:
        e = float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-30))
        worst = max(worst, e)
        bad = not torch.isfinite(got[k]).all()
        print(f"   {k:28s} rel-L2 {e:.3e}  |ref| {float(ref[k].norm()):.3e} {'NONFINITE' if bad else ''}")
    print(f"   worst {worst:.3e}")

FL = 2 * 587264
for prec in ("f32", "f16x3", "f16", "bf16"):
    net = make(); net.train_precision = prec
    Pn = 196608
    xb = torch.cat([O.embed(torch.rand(Pn, 3) * 2 - 1, 9), torch.nn.functional.normalize(torch.randn(Pn, 3), dim=-1)], -1).to(dev)
    Gb = torch.randn(Pn, 4, device=dev) * 1e-4
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True); tf = tb = 0.0
        if it >= 2: e0.record()
        out = net(xb)
        if it >= 2: e1.record()
        out.backward(Gb)
        if it >= 2:
            e2.record(); torch.cuda.synchronize(); tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        for p in net.parameters(): p.grad = None
    print(f"{prec:6s} P={Pn} fwd(save) {tf/4:.3f} ms  bwd {tb/4:.3f} ms  -> {3*Pn*FL/((tf+tb)/4)/1e9:.1f} TFLOP/s (fwd+bwd = 3x)")

# ---- against the quantised CPU model (same sign pattern) ----
from lp_reference import nerf_forward_lp
for prec, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = nerf_forward_lp(po, x.cpu(), dt)
    (o * G.cpu()).sum().backward()
    got, og = res[prec]
    print(f"== {prec} vs quantised CPU model: forward rel-L2 {float((og.cpu() - o.detach()).norm() / o.detach().norm()):.3e}")
    for k in po:
        e = float((got[k].cpu() - po[k].grad).norm() / (po[k].grad.norm() + 1e-30))
        print(f"   {k:28s} rel-L2 {e:.3e}")
