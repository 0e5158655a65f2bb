#!/usr/bin/env python3
"""Phase timeline of mlp_wgrad_f16_kernel from its own wall-clock stamps (a -DHW_TRACE variant of the library):

    SCADE_AB_FLAGS=-DHW_TRACE SCADE_AB_OUT=tools/scratch/ab_hwt python -m scade_amd.build
    SCADE_LIB=$PWD/tools/scratch/ab_hwt/libscade_hip.so python tools/probe_wgrad_f16_trace.py [P]

Per (job width, wave): mean nanoseconds per ring stage spent waiting for the stage's data (s_waitcnt + barrier),
issuing the LDS-DMAs of the stage D - 1 ahead, and in compute() - over every workgroup of ONE backward launch.
"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from scade_amd import ops, _lib
from scade_amd.ops import PARAM_ORDER

P = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
lib = _lib.load()
if not hasattr(lib, "scade_debug_hw"):
    sys.exit("this library was not built with -DHW_TRACE (see the docstring)")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
shapes = [tuple(ops.PARAM_SHAPES[k]) for k in PARAM_ORDER]
params = [((torch.rand(sh, generator=g) * 2 - 1) * (0.1 if len(sh) == 2 else 0.0)).to(dev) for sh in shapes]
pk, pk16, pkt = ops.mlp_pack(params), ops.mlp_pack_f16(params), ops.mlp_pack_t_f16(params)
x = (torch.rand(P, 60, generator=g) * 2 - 1).to(dev)
acts = torch.zeros(int(lib.scade_mlp_acts_floats(P)), device=dev)
ops.mlp_fwd_f16(pk16, x, None, None, acts=acts, rows24=True)
go = (torch.randn(P, 4, generator=g) * 1e-4).to(dev)
for _ in range(3):
    ops.mlp_bwd_f16(pk, pkt, acts, go, wgrad_f16=True)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8 * 4, dtype=np.uint64)
lib.scade_debug_hw.argtypes = [ctypes.c_void_p]
lib.scade_debug_hw.restype = ctypes.c_int
assert lib.scade_debug_hw(buf.ctypes.data) == 0
t = buf.reshape(4096, 8, 4).astype(np.float64)
ns, kw = (buf.reshape(4096, 8, 4)[:, :, 3] & 0xffffffff).astype(np.float64), buf.reshape(4096, 8, 4)[:, :, 3] >> 32
for width in (256, 64):
    m = (kw == width) & (ns > 0)
    if not m.any():
        continue
    print(f"KW = {width}: {int(m[:, 0].sum())} workgroups, {ns[m].mean():.0f} stages each; ns per stage (100 MHz stamps)")
    for w in range(8):
        mw = m[:, w]
        if not mw.any():
            continue
        per = lambda k: 10.0 * t[mw, w, k].sum() / ns[mw, w].sum()
        print(f"   wave {w}: wait {per(0):7.0f}   issue {per(1):7.0f}   compute {per(2):7.0f}   total {per(0) + per(1) + per(2):7.0f}")
