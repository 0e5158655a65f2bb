#!/usr/bin/env python3
"""Headline benchmark: rays/s of the SCADE per-ray render path (64 coarse + 128 fine
samples) on N MI355X, one process per GPU.

  python bench.py --gpus N --steps 20 --warmup 3        (N > 1: spawns its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      (the same ranks, launched for it)

A "step" is one render_rays pass (run_scade_scannet.py:581-751, perturb=0, no_grad; the
test-render configuration BASELINE.json quotes the metric on) over a batch of 1024
synthetic rays PER GPU (weak scaling; rays are independent so there is no data-path
collective in this mode).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
LP_MFMA_PEAK_TFLOPS = 2500.0           # dense bf16 / f16 MFMA peak
HBM_PEAK_GBS = 8000.0                  # HBM3E
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
    ap.add_argument("--no-rayops", action="store_true", help="skip the per-ray kernels' GB/s table")
    ap.add_argument("--no-graph", action="store_true", help="skip the eager-vs-graph A/B regions (profiling passes)")
    ap.add_argument("--secondary-budget", type=float, default=420.0,
                    help="seconds the secondary regions may take in total before the line is printed without the rest")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` as a plain command: re-exec as N ranks (one process per GPU) through
    torch.distributed.run on 127.0.0.1 - the launch line the driver itself uses.  Rank 0 of the child
    job prints the JSON line on the inherited stdout."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("SCADE_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} HIP device(s)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // max(1, args.gpus))))
    env["SCADE_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


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


def cpu_baseline(pc, pf, n_rays, n_hyp):
    """The reference path (CPU restatement, verified bit-exact against the imported
    reference) timed on this box's host cores: (i) render_rays forward,
    no_grad, perturb=0 - the headline's workload; (ii) the train step - render_rays(perturb=1) +
    3-term loss with K hypotheses + backward through autograd.  Bounded: a 128-ray probe picks the
    best thread count among a few candidates, then the 1024-ray batch is timed, 1 warm-up + best of
    <= 5, each leg inside its own ~20 s budget."""
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
        best, runs = float("inf"), 0
        deadline = time.time() + 20.0
        O.render_rays(rays, pc, pf, bbc, bbs)                  # warm-up
        for _ in range(5):
            t0 = time.perf_counter()
            O.render_rays(rays, pc, pf, bbc, bbs)
            best = min(best, time.perf_counter() - t0)
            runs += 1
            if time.time() > deadline:
                break
    # train step: forward with jitter + loss + backward (the optimizer update is negligible beside it)
    g = torch.Generator().manual_seed(1)
    tgt = torch.rand(n_rays, 3, generator=g)
    hyp = torch.rand(n_hyp, n_rays, 1, generator=g) * 4.9 + 0.1
    t_rand = torch.rand(n_rays, N_COARSE, generator=g)
    u1, u2 = torch.rand(n_rays, N_FINE, generator=g), torch.rand(n_rays, N_FINE, generator=g)
    qc = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    qf = {k: v.clone().requires_grad_(True) for k, v in pf.items()}

    def train_once():
        for q in (qc, qf):
            for v in q.values():
                v.grad = None
        ret = O.render_rays(rays, qc, qf, bbc, bbs, t_rand=t_rand, u_coarse=u1, u_fine=u2)
        O.train_loss(ret, tgt, hyp)[0].backward()

    train_once()                                               # warm-up
    tbest, truns = float("inf"), 0
    deadline = time.time() + 20.0
    for _ in range(5):
        t0 = time.perf_counter()
        train_once()
        tbest = min(tbest, time.perf_counter() - t0)
        truns += 1
        if time.time() > deadline:
            break
    return {"value": n_rays / best, "unit": "rays/s", "cores": best_thr, "kind": "port",
            "host_cores_available": avail,
            "sample": f"render_rays forward (no_grad, perturb=0) on {n_rays} synthetic rays x (64+128) "
                      f"samples, best of {runs} after 1 warm-up, torch {torch.__version__} CPU, "
                      f"{best_thr} threads (best of {cands} on a 128-ray probe)",
            "train_step": {"value": n_rays / tbest, "unit": "rays/s", "cores": best_thr,
                           "sample": f"render_rays(perturb=1) + mse + 0.007 carve(K={n_hyp}) + mse0 + backward on "
                                     f"{n_rays} rays, best of {truns} after 1 warm-up (no optimizer update)"}}


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


TRAIN_PEAKS = {"f32": (FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA"), "f16x3": (F16X3_EFFECTIVE_PEAK_TFLOPS, "f16 MFMA / 3"),
               "bf16": (LP_MFMA_PEAK_TFLOPS, "bf16 MFMA"), "f16": (LP_MFMA_PEAK_TFLOPS, "f16 MFMA"),
               "bf16-s8": (LP_MFMA_PEAK_TFLOPS, "bf16 MFMA; rows saved for the weight gradient as 8-bit e5m2")}


def train_region(args, dev, world, rank, barrier, precision="f32", rays_per_gpu=None, allreduce="single",
                 graphed=False, scaling="weak"):
    """Secondary measurement (BASELINE.json configs[2]/[3]): full train step = render_rays
    (perturb=1) + mse + 0.007*space-carving(K hypotheses) + mse0, backward, the RCCL sum-all-reduce of
    the ONE flat gradient bucket (world > 1; "overlap": in two pieces, the coarse network's behind the
    coarse backward chain), fused Adam.  ``rays_per_gpu`` rays on every rank; ``graphed``: the whole
    step (collective included) replayed as one HIP graph."""
    import torch.distributed as dist
    from scade_amd import ops, parallel
    from scade_amd.graphs import GraphedTrainer
    from scade_amd.train import Trainer, make_scade_nets
    from scade_amd.synthetic import synthetic_rays
    n = rays_per_gpu or args.rays
    coarse, fine = make_scade_nets(dev, seed=0)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=precision,
                 allreduce=allreduce)
    tr.force_allreduce = dist.is_initialized() and world == 1      # one-rank RCCL self-test of the exchange
    rays = synthetic_rays(n, seed=2000 + rank).to(dev)
    parallel.seed_rank_streams(7000)                 # identical weights above, per-rank distinct jitter / u draws
    g = torch.Generator(device="cpu").manual_seed(3000 + rank)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    hyp = (torch.rand(args.hyp, n, 1, generator=g) * 4.9 + 0.1).to(dev)
    if graphed:
        gt = GraphedTrainer(tr, n, args.hyp)
        one = lambda: gt.step(rays, tgt, hyp)
    else:
        one = lambda: tr.step(rays, tgt, hyp)[0]
    # (with a process group the first steps also pay RCCL's one-time work - channel setup, the first
    # launch of its kernels beside ours - which has been seen to leak past five warm-up steps)
    # secondary region: steady state, not the contract's W / K (those govern the headline only) - at least
    # 30 warm-up and 100 timed steps (0.1 s of a 16-bit step), so that allocator start-up and the clock ramp are not
    # in the figure and one host hiccup is 1 % of it, not 2.5 %
    nwarm, nsteps = max(30, args.warmup) + (10 if dist.is_initialized() else 0), max(100, args.steps)
    for _ in range(nwarm):
        one()
    # the figure is the whole timed span between two barriers (as before); device events at the boundaries of five
    # blocks (no host synchronisation in between) give its spread - box clock / power state moves these regions by
    # a few per cent inside one run
    nblk = 5
    per = (nsteps + nblk - 1) // nblk
    nsteps = per * nblk
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nblk + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    for b in range(nblk):
        for _ in range(per):
            loss = one()
        ev[b + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    blk = [ev[b].elapsed_time(ev[b + 1]) / per for b in range(nblk)]
    assert bool(torch.isfinite(loss))
    timer = ops.KernelTimer()        # per-kernel events in a few EXTRA steps (they cost ~1 % of a step)
    if not graphed:
        ops.KERNEL_TIMER = timer
        for _ in range(5):
            one()
        barrier()
        ops.KERNEL_TIMER = None
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    flops = 3.0 * n * (N_COARSE + N_COARSE + N_FINE) * FLOP_PER_POINT   # fwd + dgrad + wgrad
    peak, peak_name = TRAIN_PEAKS[precision]
    tfl = flops / (elapsed / nsteps) / 1e12
    out = {"value": n * world * nsteps / elapsed, "unit": "rays/s", "precision": precision,
           "ms_per_step": elapsed / nsteps * 1e3, "ms_per_step_blocks": {"median": sorted(blk)[nblk // 2], "min": min(blk), "max": max(blk), "blocks": nblk},
           "steps_timed": nsteps, "rays_per_gpu": n, "global_rays": n * world,
           "scaling": scaling, "hypotheses": args.hyp, "graphed": graphed,
           "collective": (f"{dist.get_backend()} all-reduce(sum, fp32) of ONE bucket of {tr.bucket.numel} floats per "
                          f"step over {world} rank(s), mode={allreduce}"
                          + (" (one-rank group: the collective is issued, nothing crosses a link)" if world == 1 else "")
                          if (world > 1 or tr.force_allreduce) else "none (1 GPU, no process group)"),
           "rccl_ranks": world if (world > 1 or tr.force_allreduce) else 0,
           "whole_step_tflops_per_gpu": tfl,
           "whole_step_frac_of_peak": tfl / peak, "peak": f"{peak:.1f} TFLOP/s ({peak_name})"}
    if not graphed:
        kb = timer.summary()["mlp_bwd"]
        out["mlp_bwd_tflops"] = kb["work"] / (kb["ms"] * 1e-3) / 1e12
    return out


def rayops_region(dev, n_rays=16384, n_hyp=20, iters=20):
    """The per-ray kernels ("HBM/latency-bound, reported as GB/s vs 8 TB/s") at
    the 16,384-ray chunk of a full-image render: HIP-event time per launch and ALGORITHMIC bytes = every
    input tensor read once + every output written once (a stride-0 broadcast input counts once)."""
    from scade_amd import ops
    from scade_amd.synthetic import synthetic_rays
    g = torch.Generator().manual_seed(9)
    rays = synthetic_rays(n_rays, seed=9).to(dev)
    S0, Si = N_COARSE, N_FINE
    S1 = S0 + Si
    raw0 = torch.randn(n_rays, S0, 4, generator=g).to(dev)
    raw1 = torch.randn(n_rays, S1, 4, generator=g).to(dev)
    u = torch.rand(n_rays, Si, generator=g).to(dev)
    hyp = (torch.rand(n_hyp, n_rays, 1, generator=g) * 4.9 + 0.1).to(dev)
    tgt = torch.rand(n_rays, 3, generator=g).to(dev)
    z0, _ = ops.ray_points(rays, S0, None, False)
    tail0 = ops.ray_tail(raw0, z0, rays, None, u, Si, merge=True, want_samples=False)
    z1 = tail0[7]
    tail1 = ops.ray_tail(raw1, z1, rays, None, u, Si, merge=False, want_std=True)
    w1, pred = tail1[3], tail1[5]
    gw = torch.randn(n_rays, S1, generator=g).to(dev)
    g3, g1 = torch.randn(n_rays, 3, generator=g).to(dev), torch.randn(n_rays, generator=g).to(dev)

    def nbytes(ts):
        tot = 0
        for t in ts:
            if t is None:
                continue
            tot += (t.untyped_storage().nbytes() if any(st == 0 for st in t.stride()) and t.numel() > 1
                    else t.numel() * t.element_size())
        return tot

    table = {}

    def timed(name, fn, ins):
        outs = fn()
        outs = [o for o in (outs if isinstance(outs, (tuple, list)) else [outs]) if torch.is_tensor(o)]
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        by = nbytes(ins) + nbytes(outs)
        gbs = by / (us * 1e-6) / 1e9
        table[name] = {"us": round(us, 2), "bytes_per_ray": round(by / n_rays, 1), "GBps": round(gbs, 1),
                       "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)}

    rows = rays[:, :8]
    timed("ray_points(S=64)", lambda: ops.ray_points(rays, S0, None, False), [rows])
    timed("ray_tail coarse: composite+sample_pdf+merge+points (64 -> 192)",
          lambda: ops.ray_tail(raw0, z0, rays, None, u, Si, merge=True, want_samples=False), [raw0, z0, rows, u])
    timed("ray_tail fine: composite+sample_pdf+z_std (S=192)",
          lambda: ops.ray_tail(raw1, z1, rays, None, u, Si, merge=False, want_std=True), [raw1, z1, rows, u])
    timed("composite_fwd(S=192)", lambda: ops.composite_fwd(raw1, z1, rays[:, 3:6]), [raw1, z1, rays[:, 3:6]])
    timed("composite_bwd(S=192)", lambda: ops.composite_bwd(raw1, z1, rays[:, 3:6], None, g3, None, None, gw, None),
          [raw1, z1, rays[:, 3:6], g3, gw])
    timed("sample_pdf_fwd(M=191,S=128)",
          lambda: ops.sample_pdf_fwd(z1, w1[:, 1:-1], u, Si, bins_are_mids=True)[0], [z1, w1, u])
    timed("sample_pdf_bwd(M=191,S=128)", lambda: ops.sample_pdf_bwd(z1, w1[:, 1:-1], u, u, True), [z1, w1, u, u])
    timed("merge_sorted(64+128)+points", lambda: ops.merge_sorted(z0, pred, rays), [z0, pred, rows])
    pr = pred.detach().requires_grad_(True)
    hy = hyp.detach().requires_grad_(True)

    def carve_fb():
        pr.grad = hy.grad = None
        ops.CarveFn.apply(pr, hy, None, 0.0, False).backward()
        return pr.grad, hy.grad
    with torch.no_grad():
        timed(f"carve_fwd(K={n_hyp},P=128)", lambda: ops.CarveFn.apply(pred, hyp, None, 0.0, False), [pred, hyp])
    timed(f"carve fwd+bwd (K={n_hyp},P=128, 2 launches + autograd)", carve_fb, [pred, hyp, pred, hyp])
    with torch.no_grad():
        timed("mse_fwd", lambda: ops.MseFn.apply(g3, tgt, None), [g3, tgt])
    return {"rays": n_rays, "note": "algorithmic bytes / HIP-event time per launch; peak 8 TB/s", "kernels": table}


def interleaved_ab(fns, steps, reps, warmup):
    """Same-process A/B harness: the variants in ``fns`` (name -> callable doing ONE step) are warmed up together and
    then timed in ALTERNATING blocks of ``steps`` steps, ``reps`` times each (A B A B ...), so that clock / thermal
    drift and allocator state hit every variant alike.  -> {name: {"median_ms", "min_ms", "max_ms", "blocks"}}.
    A difference between two variants is resolved when it exceeds their spreads; one block each (what the
    regions above time) is not evidence for anything below ~5 %."""
    for _ in range(warmup):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    blocks = {k: [] for k in fns}
    for _ in range(reps):
        for k, f in fns.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                f()
            torch.cuda.synchronize()
            blocks[k].append((time.perf_counter() - t0) / steps * 1e3)
    out = {}
    for k, v in blocks.items():
        sv = sorted(v)
        out[k] = {"median_ms": sv[len(sv) // 2], "min_ms": sv[0], "max_ms": sv[-1], "blocks": len(sv)}
    return out


def graph_region(args, dev, n_rays, precision, n_hyp=None):
    """Secondary measurement (single process): the same train step at ``n_rays`` rays, eager vs captured in one
    HIP graph (scade_amd/graphs.py), as an INTERLEAVED A/B (interleaved_ab: 7 alternating blocks of >= 40 steps per
    mode after a common warm-up; median and spread reported).  128 rays = the per-GPU shard of a strongly-scaled
    1024-ray batch on 8 GPUs (BASELINE.json configs[3]), where the host work of an eager step is the limit and the
    graph wins; at 1024 rays both modes are GPU-bound and the replay carries one launch more than the eager step
    (the input-staging kernel, ~5 us) plus a ~9 us bubble at every replay boundary - the graph's first node waits
    for the previous replay's completion signal, where eager launches queue behind each other without a gap
    (rocprofv3 timelines of both: profiles/r04_graph_vs_eager.txt) - so there the replay is 1-2 % SLOWER."""
    from scade_amd.graphs import GraphedTrainer
    from scade_amd.synthetic import synthetic_rays
    from scade_amd.train import Trainer, make_scade_nets
    n_hyp = n_hyp or args.hyp
    rays = synthetic_rays(n_rays, seed=4000).to(dev)
    g = torch.Generator(device="cpu").manual_seed(4001)
    tgt = torch.rand(n_rays, 3, generator=g).to(dev)
    hyp = (torch.rand(n_hyp, n_rays, 1, generator=g) * 4.9 + 0.1).to(dev)
    fns, last = {}, {}
    for mode in ("eager", "graph"):
        coarse, fine = make_scade_nets(dev, seed=0)
        tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=precision)
        if mode == "graph":
            gt = GraphedTrainer(tr, n_rays, n_hyp)
            fns[mode] = lambda gt=gt: last.__setitem__("graph", gt.step(rays, tgt, hyp))
        else:
            fns[mode] = lambda tr=tr: last.__setitem__("eager", tr.step(rays, tgt, hyp)[0])
    res = interleaved_ab(fns, steps=max(40, args.steps), reps=7, warmup=max(20, args.warmup))
    assert all(bool(torch.isfinite(v)) for v in last.values())
    out = {"rays": n_rays, "precision": precision, "hypotheses": n_hyp,
           "harness": "interleaved A/B, 7 alternating blocks per mode, median"}
    for mode, r in res.items():
        out[f"ms_per_step_{mode}"] = r["median_ms"]
        out[f"spread_{mode}_ms"] = [r["min_ms"], r["max_ms"]]
    out["graph_over_eager"] = res["graph"]["median_ms"] / res["eager"]["median_ms"]
    return out


def synthetic_scene(dev, n_hyp, Hh=468, Ww=624, n_train=18, n_test=1):
    """A ScanNet-sized scene held in memory, in the tuple the scene loaders return (scene.load_scene_scannet):
    468 x 624 pinhole views (fx = fy = 578), ``n_train`` training views + ``n_test`` test views
    on a short baseline, smooth colour ramps, a tilted depth plane, K hypotheses = the depth + noise clipped to
    [near, far] (data/load_scene.py:319-348).  The hypothesis stack (n_train x K x H x W floats: 420 MB at K = 20) is
    generated on the device."""
    import numpy as np
    yy, xx = np.meshgrid(np.linspace(0, 1, Hh), np.linspace(0, 1, Ww), indexing="ij")
    n_all = n_train + n_test
    imgs = np.stack([np.stack([xx, yy, 0.5 + 0.3 * np.sin(3 * xx + i)], -1) for i in range(n_all)]).astype(np.float32)
    dep = (1.0 + 1.5 * xx + 0.5 * yy).astype(np.float32)
    depths = np.repeat(dep[None, :, :, None], n_all, 0)
    valid = np.ones((n_all, Hh, Ww), bool)
    poses = np.repeat(np.eye(4, dtype=np.float32)[None], n_all, 0)
    poses[:, 0, 3] = np.linspace(0, 0.5, n_all)
    intr = np.repeat(np.array([[578.0, 578.0, 312.0, 234.0]], np.float32), n_all, 0)
    g = torch.Generator(device=dev).manual_seed(11)
    hyps = (torch.as_tensor(dep, device=dev)[None, None, :, :, None]
            + 0.2 * torch.randn(n_train, n_hyp, Hh, Ww, 1, device=dev, generator=g)).clamp_(0.1, 5.0)
    i_split = [np.arange(n_train), np.arange(0), np.arange(n_train, n_all), np.arange(0)]
    return (imgs, depths, valid, poses, Hh, Ww, intr, 0.1, 5.0, i_split, None, None, hyps)


def driver_loop_region(args, dev, scene_data, precision, n_rays, iters=300, warm=60, tail=50):
    """The training LOOP a user runs (scade_amd.driver.train_scene = the call sequence of the reference's train_nerf,
    run_scade_scannet.py:942-997: image pick, pixel pick, batch gather, render_hyp, three-term loss, backward, both
    optimizer steps), on the resident synthetic scene: ms per ITERATION all in, host clock, steady state (the first
    ``warm`` iterations - graph capture, allocator, clock ramp - are left out; the final test render and the
    checkpoint / image writes are outside the loop's clock as they are outside the reference's per-iteration work)."""
    import shutil
    import tempfile
    from scade_amd import driver
    import contextlib
    out = tempfile.mkdtemp(prefix="scade_bench_loop_")
    try:
        # (the image writer prints its metrics: stdout belongs to the JSON line)
        with contextlib.redirect_stdout(sys.stderr):
            res = driver.train_scene(scene_data, out, f"{precision}_{n_rays}", "synthetic",
                                     num_iterations=warm + iters + tail,
                                     N_rand=n_rays, i_weights=10 ** 9, i_print=10 ** 9, precision=precision,
                                     no_reload=True, loop_warmup=warm, tail_losses=tail, log=lambda *_: None,
                                     test_chunk=16384)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    assert res["graphed"] and res["trace"] and all(v == v for _, v in res["trace"])
    return {"ms_per_iteration": res["ms_per_iteration"], "rays": n_rays, "precision": precision,
            "iterations_timed": res["iterations_timed"], "graphed": res["graphed"],
            "rays_per_s": n_rays / (res["ms_per_iteration"] * 1e-3),
            # (the test PSNR of a 300-iteration run was printed here until round 5: it moves by 2 dB with the pixel
            # sampler alone - profiles/r05_driver_loop.txt - and said nothing about the precision; the trained-quality
            # comparison with its seed-to-seed spread is tools/convergence_parity.py -> profiles/r06_convergence.json)
            "loss_mean_last_50_iterations": res["tail_loss_mean"],
            "loss_note": f"mean of the three-term loss over the {res['tail_losses']} iterations behind the timed span",
            "host_calls_per_iteration": "np.random.choice (view) + scade_gather_batch + hipGraphLaunch"}


def dropin_region(args, dev, n_rays, precision="f32", steps=60, warm=15):
    """The path INTEGRATION.md section 2 gives a reference maintainer - nothing of this package above the public
    operators: run_scade_scannet.py:951-997 as written there (target_h = hyp * scale + shift, render_rays(perturb=1),
    img2mse + 0.007 compute_space_carving_loss + img2mse, loss.backward(), torch.optim.Adam over the 48 parameter
    tensors, a second torch.optim.Adam over the depth scales / shifts), eager.  ``precision`` = NeRF.train_precision of
    both networks (the one attribute a maintainer sets to opt into a 16-bit path): the host side of this path costs
    ~0.5 ms per step whatever the kernels take, which is what the 16-bit rows put on record."""
    import scade_amd as S
    from scade_amd.synthetic import synthetic_rays
    from scade_amd.train import make_scade_nets
    coarse, fine = make_scade_nets(dev, seed=0)
    coarse.train_precision = fine.train_precision = precision
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
    optimizer = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    scales = torch.ones(1, 1, device=dev, requires_grad=True)
    shifts = torch.zeros(1, 1, device=dev, requires_grad=True)
    optimizer_ss = torch.optim.Adam([scales, shifts], lr=1e-7)
    rays = synthetic_rays(n_rays, seed=6000).to(dev)
    g = torch.Generator(device="cpu").manual_seed(6001)
    tgt = torch.rand(n_rays, 3, generator=g).to(dev)
    hyp = (torch.rand(args.hyp, n_rays, 1, generator=g) * 4.9 + 0.1).to(dev)

    def one():
        target_h = hyp * scales[0] + shifts[0]                                              # :954
        ret = S.render_rays(rays, True, coarse, query, N_COARSE, N_importance=N_FINE, network_fine=fine,
                            perturb=1., raw_noise_std=0.)                                   # :963
        optimizer.zero_grad()
        optimizer_ss.zero_grad()
        loss = S.img2mse(ret["rgb_map"], tgt)                                               # :968
        loss = loss + 0.007 * S.compute_space_carving_loss(ret["pred_hyp"], target_h, is_joint=False, norm_p=2,
                                                           threshold=0.0)                   # :973-979
        loss = loss + S.img2mse(ret["rgb0"], tgt)                                           # :981-983
        loss.backward()                                                                     # :985
        optimizer.step()                                                                    # :993
        optimizer_ss.step()                                                                 # :997
        return loss
    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert bool(torch.isfinite(loss))
    return {"ms_per_step": ms, "rays": n_rays, "precision": precision, "rays_per_s": n_rays / (ms * 1e-3),
            "path": "public operators + loss.backward() + torch.optim.Adam x 2 (INTEGRATION.md section 2), eager"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("SCADE_BENCH_FORCE_SPAWN") == "1"):
        spawn_ranks(args)                       # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SCADE_BENCH_SHARE_GPU=1 + SCADE_BENCH_BACKEND=gloo: logic self-test of the multi-rank path on a box with
    # fewer GPUs than ranks (the ranks share devices; RCCL refuses that, gloo does not) - not a measurement
    if os.environ.get("SCADE_BENCH_SHARE_GPU") == "1":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # a one-rank RCCL group is the 1-GPU self-test of the collective path (SCADE_BENCH_FORCE_SPAWN=1
    # re-execs through torch.distributed.run first, SCADE_BENCH_FORCE_DIST=1 stays in this process)
    use_dist = world > 1 or os.environ.get("SCADE_BENCH_FORCE_DIST") == "1" or \
        os.environ.get("SCADE_BENCH_SPAWNED") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SCADE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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
    SETUP_STEPS = 12
    for _ in range(SETUP_STEPS):
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
    # one row per launch size (coarse 64 / fine 192 samples per ray), as the committed rocprofv3 per-dispatch
    # statistics (profiles/rNN_render_kernel_stats_timed.csv) list them
    by_launch = {}
    for work, r in sorted(timer.by_work("mlp_fwd_kernel").items()):
        ms = r["ms"] / r["launches"]
        by_launch[f"{int(round(work / ops.MLP_FLOP_PER_POINT))}_points"] = {
            "launches_timed": r["launches"], "avg_launch_ms": ms, "achieved": work / (ms * 1e-3) / 1e12,
            "frac": work / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    traffic, traffic_src, pmc_util, pmc_clock = None, None, None, None
    try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        import glob
        f = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")))[-1]
        pmc = json.load(open(f))["mlp_fwd_kernel"]
        traffic, pmc_util, traffic_src = pmc["hbm_bytes_per_launch"], pmc["mfma_util"], os.path.basename(f)
        pmc_clock = pmc.get("effective_clock_ghz")
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
        "setup_steps": SETUP_STEPS,
        "setup_steps_note": "untimed one-time steps run BEFORE the declared warm-up (first-launch kernel attributes, "
                            "allocator pools, clock ramp of a cold process)",
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
                     "traffic_source": (f"static: read from the committed profiles/{traffic_src} (separate rocprofv3 "
                                        "--pmc passes of this command), not measured in this run"
                                        if traffic_src else None),
                     "mfma_util_pmc": pmc_util,
                     "effective_clock_ghz_pmc": pmc_clock,
                     "effective_clock_note": ("static, same source as traffic: GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration "
                                              "over the timed launches of the profiled run; the 157.3 TFLOP/s peak assumes 2.4 GHz"
                                              if pmc_clock else None),
                     "by_launch": by_launch,
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
    # ---- secondary regions.  They never cost the headline: a failure is recorded in its place, and a
    # watchdog prints the line with what has been measured if they overrun their budget (a hung
    # collective must not turn a measured headline into an empty record).
    state = {"region": None, "done": False}
    deadline = time.time() + args.secondary_budget

    def emit():
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL's banner goes through C stdio: flush it out first
        except Exception:
            pass
        if rank == 0:
            print(json.dumps(out), flush=True)   # the JSON line is the LAST line of stdout

    def watchdog():
        while not state["done"]:
            time.sleep(1.0)
            if time.time() > deadline and not state["done"]:
                out["watchdog"] = (f"secondary regions exceeded {args.secondary_budget:.0f} s in "
                                   f"{state['region']!r}; line printed without the rest")
                emit()
                os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()

    def guarded(key, fn, *a, **k):
        state["region"] = key
        try:
            out[key] = fn(*a, **k)
        except Exception as exc:   # noqa: BLE001 - reported, not swallowed
            out[key] = {"error": f"{type(exc).__name__}: {str(exc).splitlines()[0] if str(exc) else ''}"}
            print(f"bench.py: secondary region {key} failed: {exc!r}", file=sys.stderr)

    # exact fp32 regions first, the opt-in reduced-precision regions after them (the 16-bit bursts
    # leave the chip in a different power state for a few milliseconds); every GRAPH-CAPTURED region last:
    # a capture that fails (a collective the backend cannot capture, say) leaves the stream invalidated
    # and every later launch of the process would fail with it
    tr_args = (args, dev, world, rank, barrier)
    strong = world > 1 and args.rays % world == 0
    if not args.no_train:
        guarded("train_step", train_region, *tr_args)
        if use_dist:
            guarded("train_step_staged", train_region, *tr_args, allreduce="staged")
            guarded("train_step_overlap", train_region, *tr_args, allreduce="overlap")
    if use_dist and not args.no_image:
        # ONE 468 x 624 test image with its rays split over the ranks, maps all-gathered
        def sharded_image():
            Hh, Ww = 468, 624
            intr = torch.tensor([578.0, 578.0, 312.0, 234.0], device=dev)
            c2w = torch.eye(4, device=dev)[:3, :4].contiguous()
            kw = dict(chunk=16384, c2w=c2w, near=0.1, far=5.0, use_viewdirs=True, network_fn=coarse, shard_group=True,
                      network_query_fn=query, N_samples=N_COARSE, N_importance=N_FINE, network_fine=fine, perturb=0.)
            with torch.no_grad():
                S.render(Hh, Ww, intr, **kw)
                barrier()
                t0 = time.perf_counter()
                rgb = S.render(Hh, Ww, intr, **kw)[0]
                barrier()
            dt = time.perf_counter() - t0
            assert rgb.shape == (Hh, Ww, 3) and bool(torch.isfinite(rgb).all())
            return {"ms_per_image": dt * 1e3, "rays_per_s": Hh * Ww / dt, "ranks": world,
                    "note": "one image, rays sharded over the ranks, per-pixel maps all-gathered to every rank"}
        guarded("full_image_468x624_sharded", sharded_image)
    if not args.no_rayops and rank == 0:
        guarded("per_ray_kernels_16384", rayops_region, dev, 16384, args.hyp)
    if use_dist:
        barrier()
    if not args.no_fast:
        for prec in ("f16x3", "bf16", "f16"):
            guarded("fast_path_" + prec, fast_region, args, dev, world, barrier, step, coarse, fine, prec)
        if not args.no_train:
            # forward + dgrad + wgrad on the split-precision kernels
            guarded("train_step_f16x3", train_region, *tr_args, precision="f16x3")
            # mixed precision (BASELINE config 5's bf16 MFMA path): 16-bit forward, dgrad and wgrad
            guarded("train_step_bf16", train_region, *tr_args, precision="bf16")
            # the same with the saved rows (activations, dZ) as 8-bit e5m2: half the HBM bytes of that step
            guarded("train_step_bf16_s8", train_region, *tr_args, precision="bf16-s8")
            if use_dist:
                guarded("train_step_bf16_s8_staged", train_region, *tr_args, precision="bf16-s8", allreduce="staged")
                guarded("train_step_bf16_overlap", train_region, *tr_args, precision="bf16", allreduce="overlap")
    if not args.no_train:
        # (a gloo group's collectives cannot be stream-captured: the logic self-test runs these rows eagerly)
        can_graph = not use_dist or dist.get_backend() == "nccl"
        if strong:
            # BASELINE.json configs[3]: ONE 1024-ray batch sharded over the ranks, step replayed as a graph
            guarded("train_step_strong_graph" if can_graph else "train_step_strong", train_region, *tr_args,
                    rays_per_gpu=args.rays // world, graphed=can_graph, scaling="strong")
        if not args.no_fast:
            if use_dist and can_graph:
                guarded("train_step_bf16_graph", train_region, *tr_args, precision="bf16", graphed=True)
                guarded("train_step_bf16_s8_graph", train_region, *tr_args, precision="bf16-s8", graphed=True)
                guarded("train_step_bf16_s8_staged_graph", train_region, *tr_args, precision="bf16-s8", graphed=True,
                        allreduce="staged")
            if strong:
                guarded("train_step_bf16_s8_strong_graph" if can_graph else "train_step_bf16_s8_strong", train_region,
                        *tr_args, precision="bf16-s8", rays_per_gpu=args.rays // world, graphed=can_graph,
                        scaling="strong")
            if world == 1 and not args.no_graph:
                guarded("train_step_graph", lambda: [graph_region(args, dev, 128, "f32"),
                                                     graph_region(args, dev, 128, "f16x3"),
                                                     graph_region(args, dev, 128, "bf16"),
                                                     graph_region(args, dev, args.rays, "bf16"),
                                                     graph_region(args, dev, 128, "bf16-s8"),
                                                     graph_region(args, dev, args.rays, "bf16-s8"),
                                                     # BASELINE.json configs[4]'s per-GPU shard: 4096 rays / 8, K = 40
                                                     graph_region(args, dev, 512, "bf16-s8", n_hyp=40)])

                def ceilings():
                    """ms(1024-ray step) / ms(128-ray graphed shard): what 8 GPUs could give a strongly-scaled
                    1024-ray batch (configs[3]) before one microsecond of RCCL."""
                    full = {"f32": out.get("train_step"), "f16x3": out.get("train_step_f16x3"),
                            "bf16": out.get("train_step_bf16"), "bf16-s8": out.get("train_step_bf16_s8")}
                    res = {}
                    for row in out["train_step_graph"]:
                        f = full.get(row["precision"])
                        if row["rays"] == 128 and isinstance(f, dict) and "ms_per_step" in f and args.rays == 1024:
                            res[row["precision"]] = f["ms_per_step"] / row["ms_per_step_graph"]
                    return res
                guarded("strong_scaling_ceiling_8gpu", ceilings)
        if world == 1 and not args.no_graph:
            # the loop a user runs (driver.train_scene, graph-captured step + fused batch gather) and the drop-in
            # operator path, beside the bare Trainer.step figures above
            def loops():
                rows = []
                data = synthetic_scene(dev, args.hyp)
                for prec, n in (("f32", args.rays), ("f32", 128)) + \
                        ((() if args.no_fast else (("bf16-s8", args.rays), ("bf16-s8", 128)))):
                    rows.append(driver_loop_region(args, dev, data, prec, n))
                return rows
            guarded("driver_loop", loops)
            guarded("train_step_dropin", lambda: [dropin_region(args, dev, args.rays)] + ([] if args.no_fast else [
                dropin_region(args, dev, args.rays, "bf16-s8"), dropin_region(args, dev, 128, "bf16-s8")]))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        state["region"] = "cpu_baseline"
        out["cpu_baseline"] = cpu_baseline(pc, pf, 1024, args.hyp)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if isinstance(out.get("train_step"), dict) and "value" in out["train_step"]:
            out["gpu_over_cpu_train_step"] = out["train_step"]["value"] / out["cpu_baseline"]["train_step"]["value"]
    if use_dist:
        out["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                       "version": ".".join(str(v) for v in torch.cuda.nccl.version())}
        if dist.get_backend() != "nccl":
            out["rccl"]["note"] = "NOT RCCL: logic self-test of the multi-rank path (SCADE_BENCH_BACKEND)"
        state["region"] = "final barrier"
        barrier()
    state["done"] = True
    emit()
    if use_dist:
        # (round 2 skipped the teardown here with os._exit: a process whose collectives had been graph-captured was
        # seen to abort about once in sixty runs.  The cause was the process group's watchdog thread polling an
        # event during a GLOBAL-mode stream capture - scade_amd/graphs.py _capture_mode - not the teardown.)
        sys.stdout.flush()
        sys.stderr.flush()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
