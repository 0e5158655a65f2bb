"""Inference forwards (split-precision and bf16) timed at 1, 2, 4 and 8 tiles per CU: what a second workgroup on a CU buys
(round 5: 9 %; tools/experiments/README.md).  python tools/probe_fwd_sizes.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from scade_amd import ops, _lib
from scade_amd.ops import PARAM_ORDER
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
params = [((torch.rand(tuple(ops.PARAM_SHAPES[k]), generator=g) * 2 - 1) * 0.1).to(dev) for k in PARAM_ORDER]
pk = ops.mlp_pack_f16(params)
pl = ops.mlp_pack_lp(params, True)
for P in (16384, 32768, 65536, 131072):
    x = (torch.rand(P, 60, generator=g) * 2 - 1).to(dev)
    for fn, name in ((lambda: ops.mlp_fwd_f16(pk, x, None, None), "f16x3"), (lambda: ops.mlp_fwd_lp(pl, True, x, None, None), "bf16")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        print(f"{name:6s} P={P:7d}  {us:8.1f} us   {us * 1000 / P:6.2f} ns/point   tiles/CU {P / (64 if name == 'f16x3' else 128) / 256:.2f}")
