import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scade_amd as S
from scade_amd import ops
from oracle import scade_oracle as O
dev = torch.device("cuda:0")
params = O.nerf_init(5)
g_ = torch.Generator().manual_seed(5)
for k_ in params:
    if k_.endswith(".bias"): params[k_] = 0.1 * torch.randn(params[k_].shape, generator=g_)
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
net.load_state_dict(params); net = net.to(dev)
torch.manual_seed(3)
N, Sm = 32, 64
P = N * Sm
pts = (torch.rand(N, Sm, 3) * 6 - 3).to(dev)
vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1).to(dev)
bb = torch.tensor([0.1, -0.2, 0.3, 0.2], device=dev)
acts = ops.mlp_acts_alloc(P, dev); acts.zero_()
out = ops.mlp_fwd_f16(net.packed_f16(), pts, vd, bb, acts)
acts_ref = ops.mlp_acts_alloc(P, dev); acts_ref.zero_()
out_ref = ops.mlp_fwd_points(net.packed(), pts, vd, bb, acts_ref)
torch.cuda.synchronize()
slot = P * 256
names = [f"slot{i}" for i in range(10)] + ["emb", "alpha"]
offs = [i * slot for i in range(10)] + [10 * slot, 10 * slot + P * 64, 10 * slot + P * 64 + P]
for i, n in enumerate(names):
    a, b = acts[offs[i]:offs[i + 1]], acts_ref[offs[i]:offs[i + 1]]
    print(f"{n:7s} rel diff vs exact kernel {float((a - b).norm() / (b.norm() + 1e-30)):.3e}")
moff = (offs[-1] + 1) // 2 * 2
ma = acts[moff:].view(torch.int32); mb = acts_ref[moff:].view(torch.int32)
x = (ma ^ mb)
# popcount of differing bits
diff_bits = sum(int(((x >> k) & 1).sum()) for k in range(32))
print("mask words:", ma.numel(), "differing bits:", diff_bits, "of", ma.numel() * 32)
