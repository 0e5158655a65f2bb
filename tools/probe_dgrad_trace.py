#!/usr/bin/env python3
"""Where a gemm of the 16-bit dgrad chain (format code 2) spends its cycles: core-clock stamps of sixteen consecutive
workgroups of the joint launch's third round at the phase boundaries of the nine gemms, with their HW_ID words so
that the pairs sharing a CU can be lined up.  Needs a -DDG_TRACE variant:

    SCADE_AB_FLAGS=-DDG_TRACE SCADE_AB_OUT=tools/scratch/ab_DT python -m scade_amd.build
    SCADE_LIB=tools/scratch/ab_DT/libscade_hip.so python tools/probe_dgrad_trace.py

Per gemm and wave: k-loop (16 k-blocks x 8 MFMAs of 32 cycles = 4,096 pipe cycles for 64 features x 128 points; the
CU's other workgroup shares the pipe) | wait at the barrier | epilogue (mask, pack, ds_write_b128) | second barrier."""
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
N, K = int(os.environ.get("TRACE_RAYS", "1024")), 20
coarse, fine = make_scade_nets(dev, seed=0)
tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision="bf16-s8")
rays = synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(1)
tgt = torch.rand(N, 3, device=dev)
hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
import time
nsteps = int(os.environ.get("TRACE_STEPS", "10"))
for _ in range(nsteps):
    tr.step(rays, tgt, hyp)
if os.environ.get("TRACE_RENDER"):                     # the inference forward (-DFL_SAVE=0 variants)
    import scade_amd as S
    coarse.inference_precision = fine.inference_precision = "bf16"
    with torch.no_grad():
        for _ in range(5):
            S.render_rays(rays, True, coarse, tr.query, 64, N_importance=128, network_fine=fine, perturb=0.)
torch.cuda.synchronize()
if os.environ.get("TRACE_IDLE"):                       # clocks after an idle gap: one step from a cold start
    time.sleep(float(os.environ["TRACE_IDLE"]))
    tr.step(rays, tgt, hyp)
    torch.cuda.synchronize()
lib = _lib.load()
if "--fwd" in sys.argv:
    if not hasattr(lib, "scade_debug_fl_trace"):
        sys.exit("this library was not built with -DFL_TRACE")
    fb = (ctypes.c_ulonglong * (16 * 4 * 48))()
    lib.scade_debug_fl_trace.argtypes = [ctypes.c_void_p]
    assert lib.scade_debug_fl_trace(fb) == 0
    f = np.array(fb[:], dtype=np.int64).reshape(16, 4, 48).astype(float)
    def rep(name, x):
        print(f"   {name:34s} mean {x.mean():8.0f}   min {x.min():8.0f}   max {x.max():8.0f}")
    print(f"training forward (format code 2), fine launch, 16 workgroups x 4 waves; lifetime {np.mean(f[..., 45] - f[..., 0]):.0f} cycles")
    rt = (f[..., 47] - f[..., 46]) / 100.0        # s_memrealtime: 100 MHz
    print(f"   the same lifetime on the constant 100 MHz counter: {rt.mean():.1f} us => core clock {np.mean(f[..., 45] - f[..., 0]) / rt.mean() / 1e3:.2f} GHz")
    rep("entry -> embedding tile (+barrier)", f[..., 1] - f[..., 0])
    rep("embedding rows saved, A/bias preload", f[..., 2] - f[..., 1])
    for L in range(8):
        start = f[..., 35] if L == 6 else f[..., 2 + 4 * L]
        if L == 6:
            rep("view pad (loads + LDS writes)", f[..., 35] - f[..., 26])
        rep(f"layer {L}: k-loop", f[..., 3 + 4 * L] - start)
        rep(f"layer {L}: barrier", f[..., 4 + 4 * L] - f[..., 3 + 4 * L])
        rep(f"layer {L}: epilogue + sign words", f[..., 5 + 4 * L] - f[..., 4 + 4 * L])
        rep(f"layer {L}: tile copy + barrier", f[..., 6 + 4 * L] - f[..., 5 + 4 * L])
    rep("alpha head", f[..., 36] - f[..., 34])
    rep("feature: k-loop", f[..., 37] - f[..., 36])
    rep("feature: barrier", f[..., 38] - f[..., 37])
    rep("feature: epilogue", f[..., 39] - f[..., 38])
    rep("feature: tile copy + barrier", f[..., 40] - f[..., 39])
    rep("views: k-loop", f[..., 41] - f[..., 40])
    rep("views: barrier", f[..., 42] - f[..., 41])
    rep("views: epilogue", f[..., 43] - f[..., 42])
    rep("views: tile copy + barrier", f[..., 44] - f[..., 43])
    rep("rgb head + outputs", f[..., 45] - f[..., 44])
    sys.exit(0)
if not hasattr(lib, "scade_debug_dg_trace"):
    sys.exit("this library was not built with -DDG_TRACE (see the docstring)")
buf = (ctypes.c_ulonglong * (16 * 4 * 9 * 6))()
hw = (ctypes.c_uint * 32)()
lib.scade_debug_dg_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.scade_debug_dg_trace.restype = ctypes.c_int
assert lib.scade_debug_dg_trace(buf, hw) == 0
t = np.array(buf[:], dtype=np.int64).reshape(16, 4, 9, 6)     # [wg][wave][gemm][stamp]
hw = np.array(hw[:], dtype=np.int64).reshape(16, 2)
t0 = t[:, :, 0, 0].min()
print("wg  xcc se cu simd slot   start(gemm 0)   end(gemm 8 store)")
ids = []
for w in range(16):
    h, x = hw[w]
    wave_id, simd, cu, sh, se = h & 15, (h >> 4) & 3, (h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7
    ids.append((int(x & 15), int(se), int(sh), int(cu)))
    print(f"{w:2d}  {x & 15:3d} {se:2d} {cu:2d} {simd:4d} {wave_id:4d}   {t[w, 0, 0, 0] - t0:10d}   {t[w, 0, 8, 3] - t0:10d}")
life = (t[:, :, 8, 5] - t[:, :, 0, 4]).astype(float)
heads = (t[:, :, 0, 0] - t[:, :, 0, 4]).astype(float)
print(f"workgroup lifetime (entry -> last barrier): mean {life.mean():.0f} cycles (min {life.min():.0f} max {life.max():.0f}); heads before the first gemm {heads.mean():.0f}")
hb = (ctypes.c_ulonglong * (16 * 4 * 6))()
lib.scade_debug_dg_heads.argtypes = [ctypes.c_void_p]
assert lib.scade_debug_dg_heads(hb) == 0
hd = np.array(hb[:], dtype=np.int64).reshape(16, 4, 6)
hd = np.concatenate([hd[..., :5], t[:, :, 0, 0:1]], -1)
dd = np.diff(hd, axis=-1).astype(float)
for i, n in enumerate(["entry -> scale read", "loads issued", "compute + stores", "barrier", "A prefetch"]):
    print(f"   heads: {n:20s} mean {dd[..., i].mean():8.0f}  min {dd[..., i].min():8.0f}  max {dd[..., i].max():8.0f}")
names = ["k-loop", "barrier 1", "epilogue", "(save)+bar 2"]
# stamps: 0 start, 1 k-loop end, 2 barrier passed, 3 epilogue done, 5 second barrier passed (= next start)
d = np.stack([t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 5] - t[..., 3]], -1).astype(float)
g = slice(1, 8)                                               # the seven 16-k-block gemms with a successor
print(f"cycles per gemm (gemms 1..7, all traced waves): {(t[:, :, 2:9, 0] - t[:, :, 1:8, 0]).mean():.0f}; MFMA pipe cycles per wave 4096, per SIMD (two waves) 8192")
for i, n in enumerate(names):
    x = d[:, :, g, i]
    print(f"   {n:14s} mean {x.mean():8.0f}   min {x.min():8.0f}   max {x.max():8.0f}")
# pairs on one CU: timelines of gemm 3 relative to the pair's first stamp
seen = {}
for w, k in enumerate(ids):
    seen.setdefault(k, []).append(w)
for k, ws in seen.items():
    if len(ws) < 2:
        continue
    a, b = ws[:2]
    base = min(t[a, 0, 3, 0], t[b, 0, 3, 0])
    print(f"CU {k}: wg {a} / wg {b}, wave 0, gemms 3..5 (start, k-loop end, barrier, epilogue end, next start) relative:")
    for w in (a, b):
        for gi in (3, 4, 5):
            print(f"     wg {w} gemm {gi}: " + " ".join(f"{int(t[w, 0, gi, j] - base):7d}" for j in (0, 1, 2, 3, 5)))
