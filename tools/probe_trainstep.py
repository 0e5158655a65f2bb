#!/usr/bin/env python3
"""Eager Trainer.step timing (1024 rays, K=20), exact and bf16, with and without the two-stream backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets
dev = torch.device("cuda:0")
N, K = int(os.environ.get("N", 1024)), 20
for prec in sys.argv[1:] or ["f32", "bf16"]:
    for overlap in (True, False):
        coarse, fine = make_scade_nets(dev, seed=0)
        tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision=prec, overlap_coarse=overlap)
        rays = synthetic_rays(N, seed=1).to(dev)
        torch.manual_seed(1)
        tgt = torch.rand(N, 3, device=dev); hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
        for _ in range(8): tr.step(rays, tgt, hyp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): tr.step(rays, tgt, hyp)
        torch.cuda.synchronize()
        print(f"{prec} overlap_coarse={overlap}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms / step")
