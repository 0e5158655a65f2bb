#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP forward kernel alone (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scade_amd as S
from scade_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True).to(dev)
packed = net.packed()
bb = torch.tensor([0., 0., 0., 0.2], device=dev)
FLOP_PT = 2 * 587264
for N, Sn in ((1024, 64), (1024, 192), (4096, 192), (16384, 192)):
    pts = (torch.rand(N, Sn, 3, device=dev) * 10 - 5)
    vd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    for _ in range(3):
        ops.mlp_fwd_points(packed, pts, vd, bb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.mlp_fwd_points(packed, pts, vd, bb)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    P = N * Sn
    tf = P * FLOP_PT / ms / 1e9
    print(f"P={P:8d}  {ms:8.3f} ms  {tf:7.1f} TFLOP/s  ({tf / 157.3 * 100:5.1f}% of fp32 MFMA peak)")
