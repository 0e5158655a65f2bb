#!/usr/bin/env python3
"""Times the exact MLP backward (dgrad + wgrad + reduce) at the bench launch sizes; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  SCADE_WGRAD_PTS sweeps the chunk length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd import ops
from scade_amd.train import make_scade_nets

dev = torch.device("cuda:0")
coarse, fine = make_scade_nets(dev, seed=0)
bb = torch.tensor([0., 0., 0., 0.2], device=dev)
SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 192), (1024, 64), (128, 192)]
for N, S in SIZES:
    P = N * S
    pts = torch.rand(N, S, 3, device=dev) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    acts = ops.mlp_acts_alloc(P, dev)
    ops.mlp_fwd_points(fine.packed(), pts, vd, bb, acts)
    g = torch.randn(P, 4, device=dev) * 1e-3
    for _ in range(3):
        ops.mlp_bwd(fine.packed(), fine.packed_t(), acts, g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        grad = ops.mlp_bwd(fine.packed(), fine.packed_t(), acts, g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"P={P:7d}: mlp_bwd {ms:.3f} ms  ({P * 2 * 1174528 / ms / 1e9:.1f} TFLOP/s dgrad+wgrad), chunks {ops._lib.load().scade_mlp_bwd_chunks(P)}, |grad| {float(grad.norm()):.4e}")
