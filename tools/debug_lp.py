#!/usr/bin/env python3
"""Debug aid: determinism and slot-by-slot contents of the 16-bit training workspaces (acts, dZ)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import scade_amd as S
from scade_amd import ops, _lib
from scade_amd._lib import call, ptr, stream
from oracle import scade_oracle as O

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
bf16 = prec == "bf16"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dt = torch.bfloat16 if bf16 else torch.float16
params = O.nerf_init(5)
net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
net.load_state_dict(params); net = net.to(dev); net.train_precision = prec
torch.manual_seed(9)
x = torch.cat([O.embed(torch.rand(P, 3) * 2 - 1, 9), torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)], -1).to(dev)
G = (torch.randn(P, 4) * 1e-3).to(dev)
lib = _lib.load()
runs = []
for it in range(3):
    acts = ops.mlp_acts_lp_alloc(P, dev); acts.zero_()
    out = ops.mlp_fwd_lp(net.packed_lp(bf16), bf16, x, None, None, acts)
    ws = torch.zeros(int(lib.scade_mlp_bwd_lp_workspace_bytes(P)), device=dev, dtype=torch.uint8)
    grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
    call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(bf16)), int(bf16), ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
    torch.cuda.synchronize()
    runs.append((acts.clone(), ws.clone(), grad.clone(), out.clone()))
a0, w0, g0, o0 = runs[0]
for i, (a, w, g, o) in enumerate(runs[1:], 1):
    print(f"run {i} vs 0: out equal {torch.equal(o, o0)}  acts equal {torch.equal(a, a0)}  dz+partials equal {torch.equal(w, w0)}  grad equal {torch.equal(g, g0)}")
    if not torch.equal(a, a0):
        v = a[:10 * P * 512].view(10, P, 512); v0 = a0[:10 * P * 512].view(10, P, 512)
        print("  acts slots differing:", [(s, int((v[s] != v0[s]).any(1).sum())) for s in range(10) if not torch.equal(v[s], v0[s])])
    dz = w[:10 * P * 512].view(10, P, 512); dz0 = w0[:10 * P * 512].view(10, P, 512)
    bad = [(s, int((dz[s] != dz0[s]).any(1).sum())) for s in range(10) if not torch.equal(dz[s], dz0[s])]
    print("  dZ slots differing (slot, rows):", bad)
# reference dZ of slot 9 (d feature) = dZv @ Wv[:, :256] in float from the kernel's own dZv rows
dzv = w0[:10 * P * 512].view(10, P, 256, 2).view(torch.uint8)
dzt = w0[:10 * P * 512].view(dt).view(10, P, 256).float()
Wv = params["views_linears.0.weight"].to(dev).to(dt).float()          # [128, 259]
ref = dzt[8][:, :128] @ Wv[:, :256]
got = dzt[9]
err = (got - ref).abs().max(1).values / (ref.abs().max(1).values + 1e-30)
print(f"d feature rows vs dZv @ Wv: median rel err {float(err.median()):.3e}, rows > 5%: {int((err > 0.05).sum())} of {P}")
bad = (err > 0.05).nonzero().flatten()[:20].tolist()
print("  first bad rows:", bad)
if bad:
    r = bad[0]
    cols = ((got[r] - ref[r]).abs() > 0.05 * ref[r].abs().max()).nonzero().flatten().tolist()
    print(f"  row {r}: bad columns {cols[:64]}")
