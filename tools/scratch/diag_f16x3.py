import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from oracle import scade_oracle as O
dev = torch.device("cuda:0")
params = O.nerf_init(5)
def make(prec):
    net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
    net.load_state_dict(params); net = net.to(dev); net.train_precision = prec; return net
torch.manual_seed(3)
for (N, Sm) in ((32, 64), (37, 5), (300, 7)):
    pts = (torch.rand(N, Sm, 3) * 6 - 3).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1).to(dev)
    bb = torch.tensor([0.1, -0.2, 0.3, 0.2], device=dev)
    G = (torch.randn(N, Sm, 4) * 1e-4).to(dev)
    res = {}
    for prec in ("f32", "f16x3"):
        net = make(prec)
        out = net.forward_points(pts, vd, bb)
        (out * G).sum().backward()
        res[prec] = ({k: p.grad.clone() for k, p in net.named_parameters()}, out.detach())
    print(f"N={N} S={Sm} fwd rel {float((res['f16x3'][1]-res['f32'][1]).norm()/res['f32'][1].norm()):.2e}", end="  worst grad rel: ")
    worst = max((float((res['f16x3'][0][k]-res['f32'][0][k]).norm()/(res['f32'][0][k].norm()+1e-30)), k) for k in res['f32'][0])
    print(f"{worst[0]:.2e} ({worst[1]})")
