#!/usr/bin/env python3
"""Soak run: 3000 graph-captured train steps per precision on a teacher-render problem (random 512-ray
batches of an 8192-ray pool); reports parameter finiteness, the loss trace and the eval PSNR.  This
run found the denormal-gradient scale overflow fixed in mlp_bwd_lp.hip / mlp_bwd_f16.hip."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import scade_amd as S
from scade_amd.train import Trainer, make_scade_nets
from scade_amd.graphs import GraphedTrainer
from scade_amd.synthetic import synthetic_rays
dev = torch.device("cuda:0")
N, K, steps = 512, 20, int(os.environ.get("SOAK_STEPS", "3000"))
tc, tf = make_scade_nets(dev, seed=100)
e, _ = S.get_embedder(9, 0); ed, _ = S.get_embedder(0, 0)
query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
pool = synthetic_rays(8192, seed=21).to(dev)
with torch.no_grad():
    t = S.render_rays(pool, True, tc, query, 64, N_importance=128, network_fine=tf, perturb=0.)
tgt_all = t["rgb_map"].clone(); depth_all = t["depth_map"].clone()
for prec in os.environ.get("SOAK_PRECISIONS", "f32,f16x3,bf16,f16").split(","):
    coarse, fine = make_scade_nets(dev, seed=7)
    torch.manual_seed(int(os.environ.get("SOAK_DRAW_SEED", "7")))     # the draws' stream (in-kernel key / torch.rand)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=prec)
    gt = GraphedTrainer(tr, N, K)
    g = torch.Generator(device=dev).manual_seed(5)
    bad = 0; t0 = time.time(); hist = []
    for it in range(steps):
        idx = torch.randint(0, 8192, (N,), device=dev, generator=g)
        hyp = (depth_all[idx][None, :, None] + 0.3 * torch.randn(K, N, 1, device=dev, generator=g)).clamp(0.1, 5.0)
        loss = gt.step(pool[idx], tgt_all[idx], hyp)
        if it % max(250, steps // 12) == max(250, steps // 12) - 1:
            l = float(loss); hist.append(l)
            if not (l == l) or l > 1e3: bad += 1
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(tr.flat.data).all())
    with torch.no_grad():
        coarse.inference_precision = fine.inference_precision = "f32"
        r = S.render_rays(pool[:2048], True, coarse, query, 64, N_importance=128, network_fine=fine, perturb=0.)
    psnr = float(-10 * torch.log10(torch.mean((r["rgb_map"] - tgt_all[:2048]) ** 2)))
    print(f"{prec}: {steps} graphed steps in {time.time()-t0:.1f}s, params finite {ok}, bad {bad}, eval PSNR {psnr:.2f} dB, loss trace {[round(x,5) for x in hist]}")
