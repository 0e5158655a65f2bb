#!/usr/bin/env python3
"""Throughput of the thin driver's loop (scade_amd/driver.py) on a ScanNet-sized synthetic scene held in memory
(468 x 624, 18 training views, K = 20, 1024-ray batches): ms per iteration against the bare Trainer.step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scade_amd import driver

Hh, Ww, NT, K = 468, 624, 18, 20
rng = np.random.RandomState(0)
yy, xx = np.meshgrid(np.linspace(0, 1, Hh), np.linspace(0, 1, Ww), indexing="ij")
imgs = np.stack([np.stack([xx, yy, 0.5 + 0.3 * np.sin(3 * xx + i)], -1) for i in range(NT + 1)]).astype(np.float32)
dep = (1.0 + 1.5 * xx + 0.5 * yy).astype(np.float32)
depths = np.repeat(dep[None, :, :, None], NT + 1, 0)
valid = np.ones((NT + 1, Hh, Ww), bool)
poses = np.repeat(np.eye(4, dtype=np.float32)[None], NT + 1, 0)
poses[:, 0, 3] = np.linspace(0, 0.5, NT + 1)
intr = np.repeat(np.array([[578.0, 578.0, 312.0, 234.0]], np.float32), NT + 1, 0)
hyps = np.clip(dep[None, None, :, :, None] + 0.2 * rng.randn(NT, K, Hh, Ww, 1).astype(np.float32), 0.1, 5.0)
i_split = [np.arange(NT), np.arange(0), np.arange(NT, NT + 1), np.arange(0)]
data = (imgs, depths, valid, poses, Hh, Ww, intr, 0.1, 5.0, i_split, None, None, hyps)
iters = int(os.environ.get("ITERS", "300"))
for prec in sys.argv[1:] or ["f32", "bf16-s8"]:
    for n_rand in (1024, 128):
        for sampler, graph in (("device", True), ("device", False), ("numpy", True)):
            t0 = time.time()
            res = driver.train_scene(data, "/tmp/probe_driver_ckpt", f"{prec}_{sampler}_{int(graph)}_{n_rand}", "synthetic",
                                     num_iterations=iters + 60, N_rand=n_rand, i_weights=10 ** 9, i_print=iters + 60,
                                     precision=prec, no_reload=True, pixel_sampler=sampler, graph=graph, loop_warmup=60,
                                     log=lambda *_: None)
            torch.cuda.synchronize()
            print(f"{prec:7s} {n_rand:4d} rays pixel_sampler={sampler:6s} {'graph replay' if graph else 'eager step  '}: "
                  f"{res['ms_per_iteration']:.3f} ms / iteration over {res['iterations_timed']} iterations (whole call incl. "
                  f"set-up and the test image {time.time() - t0:.2f} s), final loss {res['trace'][-1][1]:.5f}, "
                  f"test psnr {res['test']['psnr']:.2f}")
