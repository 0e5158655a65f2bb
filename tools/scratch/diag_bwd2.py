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
cap = {}
def wrap(name, orig):
    def f(*a, **k):
        g_out = a[3]
        cap.setdefault(cur[0], {})[g_out.numel() // 4] = (g_out.detach().clone(), a[2].detach().clone())
        return orig(*a, **k)
    return f
ops.mlp_bwd_f16 = wrap("f16", ops.mlp_bwd_f16)
ops.mlp_bwd = wrap("f32", ops.mlp_bwd)
cur = [None]
rets = {}
for prec in ("f32", "f16x3"):
    cur[0] = prec
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    coarse.train_precision = fine.train_precision = prec
    scale = torch.ones(1, device=dev, requires_grad=True); shift = torch.zeros(1, device=dev, requires_grad=True)
    ret, loss = train_step(dev, g, coarse, fine, query, scale, shift)
    loss.backward()
    rets[prec] = {k: v.detach().clone() for k, v in ret.items() if torch.is_tensor(v)}
for k in ("rgb0", "weights0", "depth0", "z_vals", "rgb_map", "pred_hyp"):
    a, b = rets["f16x3"][k], rets["f32"][k]
    print(f"{k:10s} rel diff {float((a-b).norm()/b.norm()):.3e}")
for P in (2048, 6144):
    ga, aa = cap["f16x3"][P]; gb, ab = cap["f32"][P]
    print(f"P={P} g_out rel diff {float((ga-gb).norm()/gb.norm()):.3e}  acts(slots) rel diff {float((aa[:10*P*256]-ab[:10*P*256]).norm()/ab[:10*P*256].norm()):.3e}")

for P in (2048, 6144):
    ga, aa = cap["f16x3"][P]; gb, ab = cap["f32"][P]
    o_emb = 10 * P * 256; o_alpha = o_emb + P * 64; o_mask = (o_alpha + P + 1) // 2 * 2
    print(f"P={P} emb rel diff {float((aa[o_emb:o_alpha]-ab[o_emb:o_alpha]).norm()/ab[o_emb:o_alpha].norm()):.3e} alpha {float((aa[o_alpha:o_alpha+P]-ab[o_alpha:o_alpha+P]).norm()/ab[o_alpha:o_alpha+P].norm()):.3e}")
    ma = aa[o_mask:].view(torch.int32); mb = ab[o_mask:].view(torch.int32)
    x = ma ^ mb
    nb = sum(int(((x >> k) & 1).sum()) for k in range(32))
    per_layer = x.view(8, -1)
    print("   mask bits differing:", nb, "per layer:", [int(sum(int(((per_layer[l] >> k) & 1).sum()) for k in range(32))) for l in range(8)])
