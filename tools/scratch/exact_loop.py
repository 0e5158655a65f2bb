import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from oracle import scade_oracle as O
dev = torch.device("cuda:0")
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
net.load_state_dict(O.nerf_init(5)); net = net.to(dev)
Pn = 196608
xb = torch.cat([O.embed(torch.rand(Pn, 3) * 2 - 1, 9), torch.nn.functional.normalize(torch.randn(Pn, 3), dim=-1)], -1).to(dev)
Gb = torch.randn(Pn, 4, device=dev) * 1e-4
for it in range(8):
    out = net(xb); out.backward(Gb)
    for p in net.parameters(): p.grad = None
torch.cuda.synchronize()
