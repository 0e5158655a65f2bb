import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from scade_amd import ops
from conftest import load_golden
from test_oracle_golden import f6_params
from test_gpu_render import build
from test_gpu_train import train_step
dev = torch.device("cuda:0")
g = load_golden("f6_render")
pc, pf = f6_params(g)
for rep in range(2):
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    coarse.train_precision = fine.train_precision = "f16x3"
    cap = {}
    orig = ops.mlp_bwd_f16
    def wrapped(packed, packed_t_f16, acts, g_out, wgrad_f16=True):
        flat = orig(packed, packed_t_f16, acts, g_out, wgrad_f16)
        cap[g_out.numel() // 4] = (acts, g_out, flat)      # references only, no extra launches
        return flat
    ops.mlp_bwd_f16 = wrapped
    scale = torch.ones(1, device=dev, requires_grad=True); shift = torch.zeros(1, device=dev, requires_grad=True)
    ret, loss = train_step(dev, g, coarse, fine, query, scale, shift)
    loss.backward()
    torch.cuda.synchronize()
    ops.mlp_bwd_f16 = orig
    for P, net in ((6144, fine), (2048, coarse)):
        acts, g_out, flat = cap[P]
        ref = ops.mlp_bwd(net.packed(), net.packed_t(), acts, g_out)
        again = orig(net.packed(), net.packed_t_f16(), acts, g_out, True)
        pg = torch.cat([p.grad.reshape(-1) for p in net.ordered_params()])
        print(f"rep {rep} P={P}: in-flow flat vs exact {float((flat-ref).norm()/ref.norm()):.3e}  re-run vs exact {float((again-ref).norm()/ref.norm()):.3e}  p.grad vs in-flow flat {float((pg-flat).norm()/ref.norm()):.3e}")
