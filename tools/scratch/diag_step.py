import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from conftest import load_golden, rel_l2
from test_oracle_golden import f6_params
from test_gpu_render import build
from test_gpu_train import train_step
dev = torch.device("cuda:0")
g = load_golden("f6_render")
pc, pf = f6_params(g)
for prec in ("f32", "f16x3", "f16x3-dgrad"):
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    coarse.train_precision = fine.train_precision = prec
    scale = torch.ones(1, device=dev, requires_grad=True); shift = torch.zeros(1, device=dev, requires_grad=True)
    ret, loss = train_step(dev, g, coarse, fine, query, scale, shift)
    loss.backward()
    def sub(x):
        f = x.flatten(); return f if f.numel() <= 4096 else f[::97]
    errs = []
    for name, net in (("coarse", coarse), ("fine", fine)):
        for k, p in net.named_parameters():
            want = g[f"grad_{name}/{k}"]
            if float(want.abs().max()) == 0: continue
            errs.append((rel_l2(sub(p.grad), want), name + "." + k))
    errs.sort(reverse=True)
    print(prec, "loss", float(loss), "worst:", [(f"{e:.2e}", n) for e, n in errs[:4]])
