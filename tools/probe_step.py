#!/usr/bin/env python3
"""80 eager 1024-ray Trainer.step calls at one precision (argv[1]: f32 | f16x3 | bf16 | f16), the last 40 timed:
the command behind tools/timeline.sh / tools/kstats.sh runs of a train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets
dev = torch.device("cuda:0")
N, K = 1024, 20
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision=prec, overlap_coarse=False)
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev); hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
for _ in range(40): tr.step(rays, tgt, hyp)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(40): tr.step(rays, tgt, hyp)
torch.cuda.synchronize()
print(f"{prec}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms / step")
