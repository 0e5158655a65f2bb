"""Per-tensor difference of the train step's gradient bucket between precisions at several batch sizes
(python tools/probe_bucket.py 1024 2048 4096): which path departs from the others, and from which size on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import scade_oracle as O
from scade_amd import ops
from scade_amd.train import Trainer
from test_gpu_ops import make_net

dev = torch.device("cuda", 0)
K5 = 40
precs = os.environ.get("PRECS", "f32,f16x3,bf16,bf16-s8").split(",")
for N5 in [int(a) for a in sys.argv[1:]] or [4096]:
    g = torch.Generator().manual_seed(158)
    rays = O.synthetic_rays(N5, seed=159)
    tgt = torch.rand(N5, 3, generator=g) * 0.3 + 0.35
    hyp = torch.rand(K5, N5, 1, generator=g) * 4.9 + 0.1
    mask = (torch.rand(N5, generator=g) > 0.1).float()
    draws = dict(t_rand=torch.rand(N5, 64, generator=g), u_coarse=torch.rand(N5, 128, generator=g),
                 cached_u=torch.rand(N5, 128, generator=g))
    pc, pf = O.nerf_init(160), O.nerf_init(161)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    dd = {k: v.to(dev) for k, v in draws.items()}
    grads, names = {}, None
    for p in precs:
        tr = Trainer(make_net(pc, dev), make_net(pf, dev), bbc, bbs, n_images=1, precision=p, mask_mode="wild",
                     scaleshift_lr=1e-5)
        tr.step(rays.to(dev), tgt.to(dev), hyp.to(dev), mask=mask.to(dev), **dd)
        grads[p] = tr.bucket.grad.detach().double().cpu()
        if names is None:
            names, o = [], 0
            for nn, net in (("coarse", tr.coarse), ("fine", tr.fine)):
                for k, q in zip(ops.PARAM_ORDER, net.ordered_params()):
                    names.append((f"{nn}.{k}", o, q.numel()))
                    o += q.numel()
            names += [("depth_scale", o, 1), ("depth_shift", o + 1, 1)]
        del tr
        torch.cuda.empty_cache()
    print(f"==== {N5} rays, K = {K5}: rel-L2 / norm ratio against {precs[0]}")
    for name, o, n in names:
        if "bias" in name and "alpha" not in name:
            continue
        x = grads[precs[0]][o:o + n]
        cells = []
        for p in precs[1:]:
            y = grads[p][o:o + n]
            cells.append(f"{p}: {float((x - y).norm() / x.norm()):.4f} x{float(y.norm() / x.norm()):.4f}")
        print(f"  {name:34s} " + "   ".join(cells))
