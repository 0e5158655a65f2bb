#!/usr/bin/env python3
"""Headline benchmark: rays/s of the SCADE per-ray render path (64 coarse + 128 fine
samples) on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one render_rays pass (run_scade_scannet.py:581-751, perturb=0, no_grad; the
test-render configuration BASELINE.json quotes the metric on) over a batch of 1024
synthetic rays PER GPU (weak scaling; rays are independent so there is no data-path
collective in this mode).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
FLOP_PER_POINT = 2 * 587264            # SURVEY.md section 8(d), unpadded
N_COARSE, N_FINE = 64, 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1024, help="rays per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary train-step measurement")
    ap.add_argument("--no-fast", action="store_true", help="skip the secondary split-precision (f16x3) measurement")
    ap.add_argument("--no-image", action="store_true", help="skip the secondary 16-chunk image-render measurement")
    ap.add_argument("--hyp", type=int, default=20, help="depth hypotheses per ray (train step)")
    return ap.parse_args()


def make_nets(dev):
    """Two random-init (Xavier, seeded) SCADE NeRFs on the device + their parameters as CPU dicts
    (the weights the cpu_baseline leg renders with)."""
    from scade_amd.train import make_scade_nets
    coarse, fine = make_scade_nets(dev, seed=0)
    pc = {k: v.detach().cpu().clone() for k, v in coarse.state_dict().items()}
    pf = {k: v.detach().cpu().clone() for k, v in fine.state_dict().items()}
    return pc, pf, coarse, fine


def host_cores():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (a 256-thread
    torch pool on a quota-limited container thrashes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(pc, pf, n_rays):
    """The reference path (CPU restatement, verified bit-exact against the imported
    reference) timed on this box's host cores: render_rays forward, no_grad, perturb=0.
    Bounded: a 128-ray probe picks the best thread count among a few candidates, then
    the 1024-ray batch is timed (best of <= 3) within a ~30 s budget."""
    from oracle import scade_oracle as O
    avail = host_cores()
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    probe = O.synthetic_rays(128, seed=0)
    cands = sorted({c for c in (8, 16, 32, 64, avail) if c <= avail} or {avail})
    best_thr, best_rate = cands[0], 0.0
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            O.render_rays(probe, pc, pf, bbc, bbs)
            t0 = time.perf_counter()
            O.render_rays(probe, pc, pf, bbc, bbs)
            r = 128 / (time.perf_counter() - t0)
            if r > best_rate:
                best_thr, best_rate = c, r
        torch.set_num_threads(best_thr)
        rays = O.synthetic_rays(n_rays, seed=0)
        best = float("inf")
        deadline = time.time() + 25.0
        O.render_rays(rays, pc, pf, bbc, bbs)                  # warm-up
        for _ in range(3):
            t0 = time.perf_counter()
            O.render_rays(rays, pc, pf, bbc, bbs)
            best = min(best, time.perf_counter() - t0)
            if time.time() > deadline:
                break
    return {"value": n_rays / best, "unit": "rays/s", "cores": best_thr, "kind": "port",
            "host_cores_available": avail,
            "sample": f"render_rays forward (no_grad, perturb=0) on {n_rays} synthetic rays x (64+128) "
                      f"samples, best of <=3 after 1 warm-up, torch {torch.__version__} CPU, "
                      f"{best_thr} threads (best of {cands} on a 128-ray probe)"}


F16X3_EFFECTIVE_PEAK_TFLOPS = 2500.0 / 3.0   # dense f16 MFMA peak / 3 MFMAs per fp32-class product


FAST_PATHS = {
    # precision: (kernel timer key, dtype note, effective MFMA peak in TFLOP/s, peak unit note)
    "f16x3": ("mlp_fwd_f16_kernel", "f16x3 split (2 fp16 planes per fp32 value, fp32 accumulate)",
              F16X3_EFFECTIVE_PEAK_TFLOPS, "TFLOP/s (algorithmic fp32-equivalent)"),
    "bf16": ("mlp_fwd_lp_kernel", "bf16 operands, fp32 accumulate (BASELINE config 5's bf16 MFMA path)",
             2500.0, "TFLOP/s"),
    "f16": ("mlp_fwd_lp_kernel", "fp16 operands, fp32 accumulate", 2500.0, "TFLOP/s"),
}


def fast_region(args, dev, world, barrier, step, coarse, fine, precision="f16x3"):
    """Secondary measurement: the SAME render step with an opt-in reduced-cost inference kernel
    (NeRF.inference_precision): "f16x3" = every fp32 value carried as two fp16 numbers, three f16
    MFMAs per product, fp32 accumulate (parity tests hold it to the same 1e-4 bar); "bf16"/"f16" =
    ordinary single-plane 16-bit operands (held to a PSNR bound, not to the parity bar)."""
    import torch.distributed as dist
    from scade_amd import ops
    kname, dtype_note, peak, peak_unit = FAST_PATHS[precision]
    coarse.inference_precision = fine.inference_precision = precision
    try:
        for _ in range(args.warmup):
            step()
        barrier()
        timer = ops.KernelTimer()
        ops.KERNEL_TIMER = timer
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        ops.KERNEL_TIMER = None
    finally:
        coarse.inference_precision = fine.inference_precision = "f32"
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k = timer.summary()[kname]
    ach = k["work"] / (k["ms"] * 1e-3) / 1e12
    return {"value": args.rays * world * args.steps / elapsed, "unit": "rays/s",
            "ms_per_step": elapsed / args.steps * 1e3, "dtype": dtype_note,
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": ach,
                         "peak": peak, "unit": peak_unit, "frac": ach / peak,
                         "avg_launch_ms": k["ms"] / k["launches"]}}


def train_region(args, dev, world, rank, barrier, precision="f32"):
    """Secondary measurement (BASELINE.json configs[2]/[3]): full train step = render_rays
    (perturb=1) + mse + 0.007*space-carving(K hypotheses) + mse0, backward, ONE RCCL
    all-reduce of the flat gradient bucket (world > 1), fused Adam.  1024 rays per GPU."""
    import torch.distributed as dist
    from scade_amd import ops
    from scade_amd.train import Trainer, make_scade_nets
    from scade_amd.synthetic import synthetic_rays
    coarse, fine = make_scade_nets(dev, seed=0)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=precision)
    rays = synthetic_rays(args.rays, seed=2000 + rank).to(dev)
    g = torch.Generator(device="cpu").manual_seed(3000 + rank)
    tgt = torch.rand(args.rays, 3, generator=g).to(dev)
    hyp = (torch.rand(args.hyp, args.rays, 1, generator=g) * 4.9 + 0.1).to(dev)
    for _ in range(max(2, args.warmup)):
        tr.step(rays, tgt, hyp)
    barrier()
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(rays, tgt, hyp)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    assert bool(torch.isfinite(loss))
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ks = timer.summary()
    kb = ks["mlp_bwd"]
    flops = 3.0 * args.rays * (N_COARSE + N_COARSE + N_FINE) * FLOP_PER_POINT   # fwd + dgrad + wgrad
    return {"value": args.rays * world * args.steps / elapsed, "unit": "rays/s", "precision": precision,
            "ms_per_step": elapsed / args.steps * 1e3, "rays_per_gpu": args.rays, "hypotheses": args.hyp,
            "collective": (f"RCCL all-reduce(sum, fp32) of {tr.flat.numel + tr.flat_ss.numel} floats per step"
                           if world > 1 else "none (1 GPU)"),
            "whole_step_tflops_per_gpu": flops / (elapsed / args.steps) / 1e12,
            "whole_step_frac_of_fp32_mfma_peak": flops / (elapsed / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "mlp_bwd_tflops": kb["work"] / (kb["ms"] * 1e-3) / 1e12}


def graph_region(args, dev, n_rays, precision):
    """Secondary measurement (single process): the same train step at ``n_rays`` rays, eager vs
    captured in one HIP graph (scade_amd/graphs.py).  128 rays = the per-GPU shard of a strongly-
    scaled 1024-ray batch on 8 GPUs (BASELINE.json configs[3]), where the ~1.7 ms of host work per
    eager step is the limit."""
    from scade_amd.graphs import GraphedTrainer
    from scade_amd.synthetic import synthetic_rays
    from scade_amd.train import Trainer, make_scade_nets
    out = {}
    for mode in ("eager", "graph"):
        coarse, fine = make_scade_nets(dev, seed=0)
        tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=precision)
        rays = synthetic_rays(n_rays, seed=4000).to(dev)
        g = torch.Generator(device="cpu").manual_seed(4001)
        tgt = torch.rand(n_rays, 3, generator=g).to(dev)
        hyp = (torch.rand(args.hyp, n_rays, 1, generator=g) * 4.9 + 0.1).to(dev)
        gt = GraphedTrainer(tr, n_rays, args.hyp) if mode == "graph" else None
        f = (lambda: gt.step(rays, tgt, hyp)) if gt else (lambda: tr.step(rays, tgt, hyp)[0])
        for _ in range(max(3, args.warmup)):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = f()
        torch.cuda.synchronize()
        out[f"ms_per_step_{mode}"] = (time.perf_counter() - t0) / args.steps * 1e3
        assert bool(torch.isfinite(loss))
    out["rays"] = n_rays
    out["precision"] = precision
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("SCADE_BENCH_FORCE_DIST") == "1"   # the latter: 1-GPU RCCL self-test
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import scade_amd as S
    from scade_amd import ops
    from scade_amd.synthetic import synthetic_rays

    pc, pf, coarse, fine = make_nets(dev)
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
    rays = synthetic_rays(args.rays, seed=1000 + rank).to(dev)        # distinct rays per rank

    def step():
        with torch.no_grad():
            return S.render_rays(rays, True, coarse, query, N_COARSE, N_importance=N_FINE,
                                 network_fine=fine, perturb=0., raw_noise_std=0.)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time setup, not part of the measurement: first-launch kernel attributes, the caching
    # allocator's pools, and the chip's clock ramp (the first ~10 steps of a cold process run 7 %
    # slower); the W warm-up steps and the K timed steps of the contract follow
    for _ in range(12):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ret = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    assert bool(torch.isfinite(ret["rgb_map"]).all())

    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    k = timer.summary()["mlp_fwd_kernel"]
    avg_ms = k["ms"] / k["launches"]
    flops_per_launch = k["work"] / k["launches"]
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
    traffic, traffic_src, pmc_util = None, None, None
    try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        import glob
        f = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")))[-1]
        pmc = json.load(open(f))["mlp_fwd_kernel"]
        traffic, pmc_util, traffic_src = pmc["hbm_bytes_per_launch"], pmc["mfma_util"], os.path.basename(f)
    except Exception:
        pass
    total_rays = args.rays * world * args.steps
    out = {
        "metric": "rays/sec (64c+128f samples)",
        "value": total_rays / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "test-render step: render_rays forward, perturb=0, 64 coarse + 128 fine "
                               "samples, two random-init 8x256 NeRFs (BASELINE.json configs[1], "
                               "synthetic rays/weights)",
                   "rays_per_gpu": args.rays, "global_rays": args.rays * world,
                   "parallelism": f"ray-sharded x{world}, no data-path collective (inference)"},
        "roofline": {"bound": "mfma", "kernel": "mlp_fwd_kernel", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 --pmc)",
                     "traffic_source": traffic_src, "mfma_util_pmc": pmc_util,
                     "launches_timed": k["launches"], "avg_launch_ms": avg_ms,
                     "flops_per_launch": flops_per_launch,
                     "note": "algorithmic 1,174,528 FLOP/point x mean points per launch "
                             f"(coarse {args.rays * N_COARSE:,} + fine {args.rays * (N_COARSE + N_FINE):,} per step), "
                             "HIP events on the launch stream"},
    }
    # secondary: the same test-render work as a 16-chunk image render, chunks pipelined over 2 streams
    big = synthetic_rays(args.rays * 16, seed=5000 + rank).to(dev)
    for ns in (() if args.no_image else (1, 2)):
        with torch.no_grad():
            S.batchify_rays(big, args.rays, True, streams=ns, network_fn=coarse, network_query_fn=query,
                            N_samples=N_COARSE, N_importance=N_FINE, network_fine=fine, perturb=0.)
        barrier()
        t0 = time.perf_counter()
        with torch.no_grad():
            S.batchify_rays(big, args.rays, True, streams=ns, network_fn=coarse, network_query_fn=query,
                            N_samples=N_COARSE, N_importance=N_FINE, network_fine=fine, perturb=0.)
        barrier()
        out.setdefault("image_render_16x1024", {})[f"rays_per_s_per_gpu_{ns}_stream"] = \
            big.shape[0] / (time.perf_counter() - t0)
    # BASELINE.json configs[1] at its real size: one 468 x 624 ScanNet-like image (292,032 rays,
    # pinhole fx = fy = 578, identity pose) through render() in
    # 16,384-ray chunks, rays generated on the device
    if not args.no_image:
        Hh, Ww = 468, 624
        intr = torch.tensor([578.0, 578.0, 312.0, 234.0], device=dev)
        c2w = torch.eye(4, device=dev)[:3, :4].contiguous()
        kw = dict(chunk=16384, c2w=c2w, near=0.1, far=5.0, use_viewdirs=True, network_fn=coarse,
                  network_query_fn=query, N_samples=N_COARSE, N_importance=N_FINE, network_fine=fine, perturb=0.)
        full = {}
        for prec in ("f32",) + (() if args.no_fast else ("bf16",)):
            coarse.inference_precision = fine.inference_precision = prec
            try:
                with torch.no_grad():
                    S.render(Hh, Ww, intr, **kw)
                    barrier()
                    t0 = time.perf_counter()
                    rgb = S.render(Hh, Ww, intr, **kw)[0]
                    barrier()
                full[f"ms_per_image_{prec}"] = (time.perf_counter() - t0) * 1e3
                full[f"rays_per_s_per_gpu_{prec}"] = Hh * Ww / (time.perf_counter() - t0)
                assert rgb.shape == (Hh, Ww, 3) and bool(torch.isfinite(rgb).all())
            finally:
                coarse.inference_precision = fine.inference_precision = "f32"
        out["full_image_468x624"] = full
    # exact fp32 regions first, the opt-in reduced-precision regions after them (the 16-bit bursts
    # leave the chip in a different power state for a few milliseconds)
    def guarded(key, fn, *a, **k):
        """secondary regions never cost the headline line: a failure is recorded in its place"""
        try:
            out[key] = fn(*a, **k)
        except Exception as exc:   # noqa: BLE001 - reported, not swallowed
            out[key] = {"error": f"{type(exc).__name__}: {exc}"}
            print(f"bench.py: secondary region {key} failed: {exc!r}", file=sys.stderr)

    if not args.no_train:
        guarded("train_step", train_region, args, dev, world, rank, barrier)
    if not args.no_fast:
        for prec in ("f16x3", "bf16", "f16"):
            guarded("fast_path_" + prec, fast_region, args, dev, world, barrier, step, coarse, fine, prec)
        if not args.no_train:
            # forward + dgrad + wgrad on the split-precision kernels
            guarded("train_step_f16x3", train_region, args, dev, world, rank, barrier, precision="f16x3")
            # mixed precision (BASELINE config 5's bf16 MFMA path): 16-bit forward, dgrad and wgrad
            guarded("train_step_bf16", train_region, args, dev, world, rank, barrier, precision="bf16")
            if world == 1:
                guarded("train_step_graph", lambda: [graph_region(args, dev, 128, "f32"),
                                                     graph_region(args, dev, 128, "bf16"),
                                                     graph_region(args, dev, args.rays, "bf16")])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pc, pf, 1024)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if use_dist:
        dist.destroy_process_group()
    # the JSON line is the LAST line of stdout: RCCL prints its version banner through C stdio, which
    # a pipe buffers until exit - flush it out first
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
