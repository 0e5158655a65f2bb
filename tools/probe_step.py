#!/usr/bin/env python3
"""80 Trainer.step calls at one precision (argv[1]: f32 | f16x3 | bf16 | f16), the last 40 timed: the command
behind tools/timeline.sh / tools/kstats.sh runs of a train step.  argv[2]: rays (default 1024); argv[3] = "graph":
the step replayed as one HIP graph (GraphedTrainer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets
dev = torch.device("cuda:0")
K = 20
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
graphed = len(sys.argv) > 3 and sys.argv[3] == "graph"
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision=prec, overlap_coarse=False)
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev); hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
step = tr.step
if graphed:
    from scade_amd.graphs import GraphedTrainer
    step = GraphedTrainer(tr, N, K).step
for _ in range(40): step(rays, tgt, hyp)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(40): step(rays, tgt, hyp)
torch.cuda.synchronize()
print(f"{prec} {N} rays{' graphed' if graphed else ''}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms / step")
