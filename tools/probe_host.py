#!/usr/bin/env python3
"""Host cost of an EAGER Trainer.step: cProfile of 300 steps at a launch size where the GPU is faster than the
host (argv[2] rays, default 128), per precision (argv[1]).  Prints the host time per step and the functions that
carry it (cumulative).  Run on the GPU box."""
import cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets
dev = torch.device("cuda:0")
K = 20
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16-s8"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision=prec, overlap_coarse=False)
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev); hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
for _ in range(60): tr.step(rays, tgt, hyp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): tr.step(rays, tgt, hyp)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{prec} {N} rays: host {(t1 - t0) / 300 * 1e3:.3f} ms / step issue, {(t2 - t0) / 300 * 1e3:.3f} ms / step complete")
pr = cProfile.Profile()
pr.enable()
for _ in range(300): tr.step(rays, tgt, hyp)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(top)
print(s.getvalue()[:12000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:7000])
