import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import scade_amd as S
from scade_amd import ops
from oracle import scade_oracle as O
dev = torch.device("cuda:0")
params = O.nerf_init(5)
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
net.load_state_dict(params); net = net.to(dev)
torch.manual_seed(9)
P = 1000
pts = torch.rand(P, 3) * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)
x = torch.cat([O.embed(pts, 9), vd], -1)
for bf16, dt in ((False, torch.float16), (True, torch.bfloat16)):
    acts = ops.mlp_acts_lp_alloc(P, dev)
    out = ops.mlp_fwd_lp(net.packed_lp(bf16), bf16, x.to(dev), None, None, acts)
    torch.cuda.synchronize()
    slots = acts[:10 * P * 256 * 2].view(dt).view(10, P, 256).float().cpu()
    q = lambda t: t.to(dt).float()
    p = params
    ptsq, vq = q(x[:, :57]), q(x[:, 57:])
    h = ptsq
    for i in range(8):
        pre = F.linear(h, q(p[f"pts_linears.{i}.weight"]), p[f"pts_linears.{i}.bias"])
        hq = q(F.relu(pre))
        d = (slots[i] - hq)
        nz = (d != 0).float().mean()
        print(f"{'bf16' if bf16 else 'f16'} layer {i}: rel-L2 {float(d.norm()/hq.norm()):.3e}  frac differing {float(nz):.3e}  max abs {float(d.abs().max()):.3e}")
        # continue from the KERNEL's activations so layers are judged independently
        h = slots[i]
        if i == 4: h = torch.cat([ptsq, h], -1)
