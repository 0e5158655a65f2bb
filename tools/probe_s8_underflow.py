"""Why does the bf16-s8 step's fine-network gradient vanish on some batches?  Captures the output gradients handed to
the joint 16-bit backward, the loss-scale maxima and the share of zero bytes in the 8-bit dZ rows."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import scade_oracle as O
from scade_amd import ops, _lib
from scade_amd.train import Trainer
from test_gpu_ops import make_net

dev = torch.device("cuda", 0)
K5 = int(os.environ.get("K", "40"))
cap = {}
orig = ops.mlp_bwd_lp2

def spy(packed_t_lp, bf16, acts, g_out, outs, after_first=None):
    g = [t.reshape(-1, 4) for t in g_out]
    P = [t.shape[0] for t in g]
    lib = _lib.load()
    ws = [torch.empty(int(lib.scade_mlp_bwd_lp2_workspace_bytes(P[i], P[1 - i])), device=g[i].device, dtype=torch.uint8) for i in range(2)]
    grads = [ops._grad_out(o, g[0].device) for o in outs]
    Pa = (ctypes.c_int * 2)(*P)
    ops.call("scade_mlp_bwd_lp2", ops._host_ptrs(packed_t_lp), int(bf16), ops._host_ptrs(acts), ops._host_ptrs(g),
             ctypes.cast(Pa, ctypes.c_void_p), ops._host_ptrs(ws), ops._host_ptrs(grads), ops.stream())
    torch.cuda.synchronize()
    cap["g"], cap["ws"], cap["P"] = [t.clone() for t in g], ws, P
ops.mlp_bwd_lp2 = spy
import scade_amd.mlp_bwd as MB
MB.ops = ops

for N5 in [int(a) for a in sys.argv[1:]] or [1024]:
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
    tr = Trainer(make_net(pc, dev), make_net(pf, dev), bbc, bbs, n_images=1, precision="bf16-s8", mask_mode="wild", scaleshift_lr=1e-5)
    tr.step(rays.to(dev), tgt.to(dev), hyp.to(dev), mask=mask.to(dev), **dd)
    print(f"==== {N5} rays K={K5}")
    for i, name in enumerate(("net0", "net1")):
        gg = cap["g"][i].abs().double()
        P = cap["P"][i]
        fin = torch.isfinite(gg)
        q = torch.quantile(gg[fin].flatten()[:: max(1, gg.numel() // 4_000_000)], torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], dtype=torch.float64, device=dev))
        per_pt = gg.amax(dim=1)
        top = torch.topk(per_pt, 5)
        print(f"  {name}: P={P}  nonfinite {int((~fin).sum())}  max {float(gg[fin].max()):.3e}  quantiles(50,90,99,99.9,99.99) "
              + " ".join(f"{float(v):.2e}" for v in q))
        print(f"      top points {[int(t) for t in top.indices]} (ray {[int(t) // (64 if P == N5 * 64 else 192) for t in top.indices]}) values {[f'{float(v):.2e}' for v in top.values]}")
        print(f"      g rows of the top point {cap['g'][i][int(top.indices[0])].tolist()}")
        ws = cap["ws"][i]
        dzb = ops.lp_dz_bytes(P) if hasattr(ops, "lp_dz_bytes") else None
        for slot in (7, 3, 0):
            rows = ws[slot * P * 512: slot * P * 512 + P * 256]
            print(f"      dZ8 slot {slot}: zero bytes {float((rows == 0).float().mean()) * 100:.2f} %   (+/-0: {float(((rows & 0x7f) == 0).float().mean()) * 100:.2f} %)")
