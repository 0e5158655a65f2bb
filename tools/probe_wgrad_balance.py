#!/usr/bin/env python3
"""Per-workgroup timing of the balanced 16-bit weight-gradient launch (mlp_wgrad_lp_kernel), from the kernel's own
wall-clock stamps.  Needs a -DWL_DBG variant of the library:

    SCADE_AB_FLAGS=-DWL_DBG SCADE_AB_OUT=tools/scratch/ab_DBG python -m scade_amd.build
    SCADE_LIB=tools/scratch/ab_DBG/libscade_hip.so python tools/probe_wgrad_balance.py bf16 [rays]

Prints the spread of the workgroups' end times (the launch lasts as long as its slowest workgroup) and, from the
workgroups that stayed inside ONE (network, job) entry, the cost of a 32-point stage per job relative to a
256 x 256 layer's = 16: the table behind lp_job_weights() in mlp_bwd_lp.hip."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scade_amd import _lib
from scade_amd.synthetic import synthetic_rays
from scade_amd.train import Trainer, make_scade_nets

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = 20
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision=prec, overlap_coarse=False)
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev)
hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
for _ in range(20):
    tr.step(rays, tgt, hyp)
torch.cuda.synchronize()
lib = _lib.load()
if not hasattr(lib, "scade_debug_wl"):
    sys.exit("this library was not built with -DWL_DBG (see the docstring)")
buf = (ctypes.c_ulonglong * 4096)()
lib.scade_debug_wl.argtypes = [ctypes.c_void_p]
lib.scade_debug_wl.restype = ctypes.c_int
assert lib.scade_debug_wl(buf) == 0
a2 = np.array(buf[2048:], dtype=np.int64).reshape(512, 4)
a = np.array(buf[:2048], dtype=np.int64).reshape(512, 4)
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
start, end = (a[:, 0] - t0) / 100.0, (a[:, 3] - t0) / 100.0          # us (100 MHz wall clock)
dur = end - start
print(f"{prec} {N} rays: {len(a)} workgroups, start spread {start.max():.1f} us; "
      f"end min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f} us")
names = ["l1", "l2", "l3", "l4", "l5", "l6", "l7", "feat", "views", "emb0", "emb5", "embv", "rgb"]
cost = {}
for w in range(len(a)):
    ent = int(a[w, 2])                     # entries visited, as base-100 digits of (entry + 1)
    if 0 < ent < 100 and a[w, 1] > 0:
        cost.setdefault((ent - 1) % 13, []).append(dur[w] / int(a[w, 1]))
if cost:
    heavy = np.mean([np.mean(cost[j]) for j in range(7) if j in cost])
    print("  us per stage:", {names[j]: round(float(np.mean(v)), 3) for j, v in sorted(cost.items())})
    print("  relative to a layer = 16:", {names[j]: round(float(16 * np.mean(v) / heavy), 1) for j, v in sorted(cost.items())})
order = np.argsort(dur)
print("  slowest (workgroup, us, stages, entries):", [(int(i), int(dur[i]), int(a[i, 1]), int(a[i, 2])) for i in order[-8:]])

# format code 2's ring jobs: loop time per stage without prologue / epilogue, and the epilogue (partial-row stores)
b = a2[(a2[:, 2] > 20) & (a2[:, 1] > a2[:, 0])]
if len(b):
    per = (b[:, 1] - b[:, 0]) / 100.0 / b[:, 2]
    epi = (b[:, 3] - b[:, 1]) / 100.0
    print(f"  ring jobs (last one per workgroup, {len(b)} workgroups): loop us per stage min {per.min():.3f} median {np.median(per):.3f} "
          f"max {per.max():.3f};  epilogue us median {np.median(epi):.1f} max {epi.max():.1f}")

