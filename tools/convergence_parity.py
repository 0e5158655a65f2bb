#!/usr/bin/env python3
"""Trained-quality parity of the low-precision train steps WITH a noise floor.

The scene of bench.py's ``driver_loop`` region (468 x 624, 18 training views, K = 20 hypotheses; here with
``--test-views`` held-out views) is trained for ``--iters`` iterations through ``scade_amd.driver.train_scene`` -
the loop a user runs: view pick, pixel pick, fused batch gather, graph-replayed step - once per (precision, seed).
The seed moves everything the reference's seed moves (run_scade_scannet.py:831-833): weight init, view order,
pixel permutation, the jitter / u draws.  Reported per run: the mean PSNR over the test views and the mean loss of
the last 50 iterations.  The exact fp32 runs over the seeds give the noise floor (mean, standard deviation, range);
a low-precision path "falls inside" when the mean of its runs lies within the fp32 runs' range widened by one
standard deviation, and seed by seed its difference to the fp32 run of the SAME seed is reported beside the
fp32 seed-to-seed differences.

    python tools/convergence_parity.py --iters 5000 --seeds 5 --out profiles/r06_convergence.json
"""
import argparse
import contextlib
import json
import os
import shutil
import statistics
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(v):
    return {"mean": statistics.fmean(v), "stdev": statistics.stdev(v) if len(v) > 1 else 0.0, "min": min(v), "max": max(v),
            "n": len(v)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--hyp", type=int, default=20)
    ap.add_argument("--test-views", type=int, default=4)
    ap.add_argument("--precisions", default="f32,f16x3,bf16-s8")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_convergence.json"))
    a = ap.parse_args()
    import bench
    from scade_amd import driver
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    data = bench.synthetic_scene(dev, a.hyp, n_test=a.test_views)
    precs = a.precisions.split(",")
    runs = {p: [] for p in precs}
    for seed in range(a.seeds):
        for p in precs:
            out = tempfile.mkdtemp(prefix="scade_conv_")
            t0 = time.time()
            try:
                with contextlib.redirect_stdout(sys.stderr):
                    res = driver.train_scene(data, out, f"{p}_{seed}", "synthetic", num_iterations=a.iters, N_rand=a.rays,
                                             i_weights=10 ** 9, i_print=10 ** 9, precision=p, seed=seed, no_reload=True,
                                             tail_losses=50, log=lambda *_: None, test_chunk=16384)
            finally:
                shutil.rmtree(out, ignore_errors=True)
            r = {"seed": seed, "test_psnr": float(res["test"]["psnr"]), "loss_mean_last_50": res["tail_loss_mean"],
                 "ms_per_iteration": res["ms_per_iteration"], "wall_s": time.time() - t0}
            runs[p].append(r)
            print(f"{p:8s} seed {seed}: test PSNR {r['test_psnr']:.3f} dB, loss(last 50) {r['loss_mean_last_50']:.5f}, "
                  f"{r['ms_per_iteration']:.3f} ms / iteration", file=sys.stderr, flush=True)
    summary = {}
    ref = runs.get("f32")
    floor = None
    if ref:
        ps = [r["test_psnr"] for r in ref]
        floor = stats(ps)
        floor["pairwise_abs_diff_max"] = max(abs(x - y) for x in ps for y in ps)
        summary["f32"] = {"test_psnr": floor, "loss_mean_last_50": stats([r["loss_mean_last_50"] for r in ref])}
    for p in precs:
        if p == "f32":
            continue
        ps = [r["test_psnr"] for r in runs[p]]
        s = {"test_psnr": stats(ps), "loss_mean_last_50": stats([r["loss_mean_last_50"] for r in runs[p]])}
        if floor:
            lo, hi = floor["min"] - floor["stdev"], floor["max"] + floor["stdev"]
            s["same_seed_diff_to_f32_db"] = [x - r["test_psnr"] for x, r in zip(ps, ref)]
            s["mean_diff_to_f32_db"] = s["test_psnr"]["mean"] - floor["mean"]
            s["inside_f32_seed_spread"] = bool(lo <= s["test_psnr"]["mean"] <= hi)
            s["band_db"] = [lo, hi]
        summary[p] = s
    doc = {"scene": f"bench.synthetic_scene: 468 x 624, 18 training views, {a.test_views} test views, K = {a.hyp}",
           "loop": "scade_amd.driver.train_scene (graph-replayed step, device pixel sampler)", "iterations": a.iters,
           "rays_per_iteration": a.rays, "seeds": list(range(a.seeds)), "device": torch.cuda.get_device_name(0),
           "criterion": "a precision falls inside when the mean test PSNR of its runs lies within [min - stdev, max + stdev] "
                        "of the exact fp32 runs over the same seeds",
           "summary": summary, "runs": runs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
