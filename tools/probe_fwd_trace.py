#!/usr/bin/env python3
"""Where a layer of the exact fp32 forward spends its cycles: core-clock stamps (s_memtime) taken by two workgroups
of the launch's third round at the phase boundaries of layers 1..4, for the INFERENCE kernel (mlp_fwd_kernel<1,false,2>)
and the TRAINING one (<1,true,2>: ReLU sign words + the per-wave tile copy to HBM).  Needs a -DFWD_TRACE variant:

    SCADE_AB_FLAGS=-DFWD_TRACE SCADE_AB_OUT=tools/scratch/ab_FT python -m scade_amd.build
    SCADE_LIB=tools/scratch/ab_FT/libscade_hip.so python tools/probe_fwd_trace.py

Phases per layer and wave: k-loop (32 k-blocks x 16 MFMAs of 64 cycles = 32,768 pipe cycles for the wave's 64 features x
64 points; the CU's other workgroup shares the pipe) | wait at the barrier | epilogue (bias, ReLU, sign bits,
ds_write_b128) | sign words + tile copy (training) | wait at the second barrier."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scade_amd as S
from scade_amd import _lib
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets

dev = torch.device("cuda:0")
N, K = 1024, 20
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2))
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev)
hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
for _ in range(10):
    tr.step(rays, tgt, hyp)                                   # training forward (fine launch: 3072 workgroups)
with torch.no_grad():
    for _ in range(10):
        S.render_rays(rays, True, coarse, tr.query, 64, N_importance=128, network_fine=fine, perturb=0.)
torch.cuda.synchronize()
lib = _lib.load()
if not hasattr(lib, "scade_debug_fwd_trace"):
    sys.exit("this library was not built with -DFWD_TRACE (see the docstring)")
buf = (ctypes.c_ulonglong * (4 * 4 * 24 + 64))()
lib.scade_debug_fwd_trace.argtypes = [ctypes.c_void_p]
lib.scade_debug_fwd_trace.restype = ctypes.c_int
assert lib.scade_debug_fwd_trace(buf) == 0
t = np.array(buf[:384], dtype=np.int64).reshape(4, 4, 4, 6)
sect = np.array(buf[384:], dtype=np.int64).reshape(4, 16)      # [kernel x wg][wave][layer][stamp]
names = ["k-loop", "barrier 1", "epilogue", "words+copy", "barrier 2"]
for ki, kn in ((0, "inference"), (2, "training")):
    d = np.diff(t[ki:ki + 2], axis=-1).astype(float)            # [wg][wave][layer][5 phases]
    nxt = (t[ki:ki + 2, :, 1:, 0] - t[ki:ki + 2, :, :-1, 5]).astype(float)   # bias load etc. between two layers
    per_layer = (t[ki:ki + 2, :, 1:, 0] - t[ki:ki + 2, :, :-1, 0]).astype(float)
    print(f"{kn}: cycles per layer {per_layer.mean():.0f} (min {per_layer.min():.0f} max {per_layer.max():.0f}); "
          f"MFMA pipe cycles of the layer for this wave 32768, for the SIMD's two waves 65536")
    for i, n in enumerate(names):
        x = d[..., i]
        print(f"   {n:12s} mean {x.mean():8.0f}   min {x.min():8.0f}   max {x.max():8.0f}")
    print(f"   {'between':12s} mean {nxt.mean():8.0f}")
    for wg in range(2):
        for w in range(4):
            print(f"      wg {wg} wave {w} layer 2: " + " ".join(f"{int(v):7d}" for v in d[wg, w, 1]))

sn = ["prologue", "layer 0", "layer 1", "layer 2", "layer 3", "layer 4", "layer 5", "layer 6", "layer 7", "alpha head",
      "feature", "views", "rgb head"]
print("sections (wave 0 of each traced workgroup; counts):")
print("   " + " ".join(f"{n[:9]:>9s}" for n in sn) + "     total")
for ki, kn in ((0, "inference"), (1, "inference"), (2, "training"), (3, "training")):
    d = np.diff(sect[ki, :14])
    print(f"   " + " ".join(f"{int(v):9d}" for v in d) + f"  {int(d.sum()):8d}  {kn}")
