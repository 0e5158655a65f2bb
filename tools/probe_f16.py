#!/usr/bin/env python3
"""Accuracy + speed of the split-precision (f16x3) forward kernel vs the exact fp32 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import scade_amd as S
from scade_amd import ops
from oracle import scade_oracle as O

dev = torch.device("cuda:0")
params = O.nerf_init(0)
g = torch.Generator().manual_seed(5)
for k in params:
    if k.endswith(".bias"):
        params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
net.load_state_dict(params); net = net.to(dev)
torch.manual_seed(1)
N, Sn = 64, 33
pts = torch.rand(N, Sn, 3) * 10 - 5
vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
bbc, bbs = torch.zeros(3), torch.tensor(0.2)
want = O.run_network(pts.double(), vd.double(), lambda e: O.nerf_forward({k: v.double() for k, v in params.items()}, e), bbc.double(), bbs.double())
want32 = O.run_network(pts, vd, lambda e: O.nerf_forward(params, e), bbc, bbs)
bb = torch.tensor([0., 0., 0., 0.2], device=dev)
with torch.no_grad():
    exact = net.forward_points(pts.to(dev), vd.to(dev), bb).cpu()
    net.inference_precision = "f16x3"
    fast = net.forward_points(pts.to(dev), vd.to(dev), bb).cpu()
def stats(name, a, ref):
    d = (a.double() - ref.double()).abs()
    rel = d / (ref.double().abs() + 1e-3)
    print(f"{name:28s} max abs {d.max():.3e}  rel-L2 {float(d.norm()/ref.double().norm()):.3e}  max rel(|ref|+1e-3) {rel.max():.3e}")
stats("cpu fp32 oracle vs fp64", want32, want)
stats("exact fp32 kernel vs fp64", exact, want)
stats("f16x3 kernel vs fp64", fast, want)
stats("f16x3 vs exact kernel", fast, exact)
with torch.no_grad():
    for prec in ("f16", "bf16"):
        net.inference_precision = prec
        lp = net.forward_points(pts.to(dev), vd.to(dev), bb).cpu()
        stats(prec + " kernel vs fp64", lp, want)
        xe = torch.cat([O.embed((pts.reshape(-1, 3) - bbc) * bbs, 9), vd[:, None, :].expand(N, Sn, 3).reshape(-1, 3)], -1)
        lp0 = net(xe.to(dev)).cpu().reshape(N, Sn, 4)
        stats(prec + " mode0 vs mode1", lp0, lp)
bad = ((fast.double() - want32.double()).abs() > 1e-4 * want32.double().abs() + 1e-5).sum()
print("elements outside rtol 1e-4 + atol 1e-5 vs the fp32 oracle:", int(bad), "of", fast.numel())

FLOP_PT = 2 * 587264
for prec in ("f32", "f16x3", "f16", "bf16"):
    net.inference_precision = prec
    for Nn, Ss in ((1024, 64), (1024, 192), (4096, 192)):
        p = torch.rand(Nn, Ss, 3, device=dev) * 10 - 5
        v = torch.nn.functional.normalize(torch.randn(Nn, 3, device=dev), dim=-1)
        with torch.no_grad():
            for _ in range(3): net.forward_points(p, v, bb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): net.forward_points(p, v, bb)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{prec:6s} P={Nn*Ss:8d} {ms:8.3f} ms  {Nn*Ss*FLOP_PT/ms/1e9:7.1f} algorithmic TFLOP/s")
