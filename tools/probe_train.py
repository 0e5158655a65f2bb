#!/usr/bin/env python3
"""Time one full train step (fwd + 3-term loss + bwd) and its MLP backward kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scade_amd as S
from scade_amd import ops
from oracle import scade_oracle as O

dev = torch.device("cuda:0")
N, K = 1024, 20
def mk(p):
    net = S.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, use_viewdirs=True)
    net.load_state_dict(p); return net.to(dev)
coarse, fine = mk(O.nerf_init(0)), mk(O.nerf_init(1))
PREC = sys.argv[1] if len(sys.argv) > 1 else "f32"
coarse.train_precision = fine.train_precision = PREC
e, _ = S.get_embedder(9, 0); ed, _ = S.get_embedder(0, 0)
query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
rays = O.synthetic_rays(N, seed=0).to(dev)
tgt = torch.rand(N, 3, device=dev); hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
params = list(coarse.parameters()) + list(fine.parameters())
opt = torch.optim.Adam(params, lr=5e-4)

def step():
    opt.zero_grad(set_to_none=True)
    ret = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine, perturb=1.)
    loss = S.img2mse(ret["rgb_map"], tgt) + 0.007 * S.compute_space_carving_loss(ret["pred_hyp"], hyp) \
        + S.img2mse(ret["rgb0"], tgt)
    loss.backward()
    opt.step()
    return loss

for _ in range(3): step()
torch.cuda.synchronize()
timer = ops.KernelTimer(); ops.KERNEL_TIMER = timer
t0 = time.perf_counter(); reps = 10
for _ in range(reps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
ops.KERNEL_TIMER = None
print(f"[{PREC}] train step: {dt*1e3:.3f} ms  -> {N/dt:.0f} rays/s   loss {float(l):.5f}")
for k, v in timer.summary().items():
    ms = v['ms'] / v['launches']
    print(f"  {k:16s} launches {v['launches']:3d}  avg {ms:.3f} ms  {v['work']/v['launches']/ms/1e9:.1f} TFLOP/s")
FLOP = 3 * N * 256 * 2 * 587264
print(f"  whole step algorithmic {FLOP/dt/1e12:.1f} TFLOP/s ({FLOP/dt/1e12/157.3*100:.1f}% of fp32 MFMA peak)")
