#!/usr/bin/env python3
"""Launch time of scade_ray_tail_train (fine tail + train loss + both tails' backward, one launch + the loss's
one-workgroup reduce) at 128 / 1024 / 4096 rays: HIP events around back-to-back launches on one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd import ops
from scade_amd.synthetic import synthetic_rays

dev = torch.device("cuda")
S0, Si = 64, 128
S = S0 + Si
SIZES = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(128, 20), (1024, 20), (512, 40), (4096, 40)]
for N, K in SIZES:
    g = torch.Generator().manual_seed(N)
    rays = synthetic_rays(N, seed=N).to(dev)
    raw = torch.randn(N, S, 4, generator=g).to(dev)
    raw0 = torch.randn(N, S0, 4, generator=g).to(dev)
    z0, _ = ops.ray_points(rays, S0, None, False)
    u = torch.rand(N, Si, generator=g).to(dev)
    tail0 = ops.ray_tail(raw0, z0, rays, None, u, Si, merge=True, want_samples=False)
    z1, rgb0 = tail0[7], tail0[0]
    tgt = torch.rand(N, 3, generator=g).to(dev)
    hyp = (torch.rand(K, N, 1, generator=g) * 4.9 + 0.1).to(dev)
    sc = torch.ones(1, 1, device=dev, requires_grad=True)
    sh = torch.zeros(1, 1, device=dev, requires_grad=True)
    sc.grad, sh.grad = torch.zeros_like(sc), torch.zeros_like(sh)
    one = torch.ones((), device=dev)
    fn = lambda: ops.FineTailLossFn.apply(raw, z1, rays, u, Si, raw0, z0, rgb0, tgt, hyp, sc, sh, 0, None, False, True,
                                          0.007, 0.0, 1.0, one)
    with torch.no_grad():
        for _ in range(20):
            fn()
        ops.KERNEL_TIMER = t = ops.KernelTimer()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        ops.KERNEL_TIMER = None
    k = t.summary()["ray_tail_train"]
    print(f"N={N:5d} K={K}: {k['ms'] / k['launches'] * 1e3:7.2f} us per launch (kernel + loss reduce), {k['launches']} launches")
