#!/usr/bin/env python3
"""RCCL teardown abort: how often does a one-rank process that captured all-reduces in HIP
graphs die in teardown, by teardown variant?  Parent: runs every variant N times as child processes and counts
exit codes.  Child (argv[1] = variant): a small Trainer, eager + graph-captured steps with the gradient
all-reduce forced, then the variant's teardown.

  A  dist.destroy_process_group() with the CUDAGraph objects still alive            (round 2's failing sequence)
  B  graphs destroyed (del + gc) and the device synchronised BEFORE destroy_process_group()
  C  no graph capture at all: eager all-reduces only, then destroy_process_group()
  D  graphs alive, NO destroy_process_group(): plain interpreter exit
  E  as B, then os._exit(0) right after destroy_process_group() (skip interpreter finalisation)
  G  SIX captures per run in torch's default GLOBAL capture-error mode (graphs.CAPTURE_ERROR_MODE = "global"), each
     behind eager all-reduces whose events the process group's watchdog thread is still polling
  T  the same six captures in THREAD-LOCAL capture-error mode (GraphedTrainer's default since round 3)
Round-3 finding (profiles/r03_rccl_teardown.txt): the abort is not in teardown at all.  It is the NCCL watchdog
THREAD of the process group calling hipEventQuery on the eager collectives' events while the main thread has a
GLOBAL-mode stream capture open: HIP refuses the call, the watchdog throws
(ProcessGroupNCCL.cpp Watchdog::run), std::terminate aborts the process - before "WORK_DONE" is even printed.
"""
import os, subprocess, sys, time

def child(variant, port):
    import gc
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scade_amd.graphs import GraphedTrainer
    from scade_amd.synthetic import synthetic_rays
    from scade_amd.train import Trainer, make_scade_nets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    N, K = 96, 10
    rays = synthetic_rays(N, seed=1).to(dev)
    tgt = torch.rand(N, 3, device=dev)
    hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
    coarse, fine = make_scade_nets(dev, seed=4)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), precision="bf16")
    tr.force_allreduce = True
    for _ in range(3):
        tr.step(rays, tgt, hyp)
    gt = None
    if variant in ("G", "T"):
        from scade_amd import graphs
        graphs.CAPTURE_ERROR_MODE = "global" if variant == "G" else "thread_local"
        for rep in range(6):
            for _ in range(4):                  # eager collectives: work objects for the watchdog to poll
                tr.step(rays, tgt, hyp)
            gt = GraphedTrainer(tr, N, K, force_allreduce=True)
            for _ in range(2):
                gt.step(rays, tgt, hyp)
    elif variant != "C":
        gt = GraphedTrainer(tr, N, K, force_allreduce=True)
        for _ in range(5):
            gt.step(rays, tgt, hyp)
    torch.cuda.synchronize()
    print("WORK_DONE", flush=True)
    if variant in ("B", "E"):
        gt.graph = None
        del gt
        gc.collect()
        torch.cuda.synchronize()
    if variant != "D":
        dist.destroy_process_group()
    print("TEARDOWN_DONE", flush=True)
    if variant == "E":
        sys.stdout.flush()
        os._exit(0)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    variants = sys.argv[2] if len(sys.argv) > 2 else "ABCD"
    res = {}
    port = 29600
    for v in variants:
        bad = []
        t0 = time.time()
        for i in range(n):
            port += 1
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", v, str(port)], capture_output=True, text=True, timeout=300)
            ok = p.returncode == 0 and "TEARDOWN_DONE" in p.stdout
            if not ok:
                bad.append({"run": i, "rc": p.returncode, "work_done": "WORK_DONE" in p.stdout,
                            "teardown_done": "TEARDOWN_DONE" in p.stdout, "stderr_head": [l[:300] for l in p.stderr.strip().splitlines() if "amdgpu.ids" not in l][:12]})
        res[v] = {"runs": n, "failed": len(bad), "seconds": round(time.time() - t0, 1), "failures": bad}
        print(v, res[v]["runs"], "runs", res[v]["failed"], "failed", res[v]["seconds"], "s", flush=True)
    import json
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        main()
