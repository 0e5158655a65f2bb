#!/usr/bin/env python3
"""Race screen for the LDS-DMA weight gradient: the backward of one workspace repeated under changing memory
load (a side stream streaming copies of varying size) must be BIT-identical every time - the kernels are
deterministic, so any difference is a slot read before its DMA landed or refilled before it was read out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd import ops
from scade_amd.train import make_scade_nets

dev = torch.device("cuda:0")
coarse, fine = make_scade_nets(dev, seed=0)
bb = torch.tensor([0., 0., 0., 0.2], device=dev)
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
reps = int(os.environ.get("REPS", 40))
bad = 0
for P in [int(a) for a in sys.argv[1:]] or [196608, 65536, 24576, 4097, 1001, 257]:
    N = 1
    pts = torch.rand(P, 1, 3, device=dev) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
    acts = ops.mlp_acts_alloc(P, dev)
    ops.mlp_fwd_points(fine.packed(), pts, vd, bb, acts)
    g = torch.randn(P, 4, device=dev) * 1e-3
    ref = ops.mlp_bwd(fine.packed(), fine.packed_t(), acts, g).clone()
    torch.cuda.synchronize()
    for i in range(reps):
        with torch.cuda.stream(side):
            n = (1 + (i * 37) % 8) * (16 << 20)
            big[:n].copy_(big[n:2 * n])
        got = ops.mlp_bwd(fine.packed(), fine.packed_t(), acts, g)
        if not torch.equal(got, ref):
            bad += 1
            print(f"P={P} rep {i}: max |diff| {float((got - ref).abs().max()):.3e}")
    torch.cuda.synchronize()
    print(f"P={P}: {reps} repetitions, |grad| {float(ref.norm()):.6e}, finite {bool(torch.isfinite(ref).all())}")
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
