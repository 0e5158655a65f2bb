"""Ray-sharded train step ON THE HIP KERNELS, world_size 2 (replaces the
reference's nn.DataParallel, run_scade_scannet.py:438/:455/:466).

With two or more GPUs the ranks use RCCL (backend "nccl"), one process per GPU, and the graphed
sharded step is covered too.  On a one-GPU box both ranks share cuda:0 and the collective runs over
gloo (RCCL refuses two ranks on one device): the kernels, the sharding, the share weighting, the
single bucket, the two-piece overlapped exchange and the is_joint exchange are the same code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_l2

pytestmark = pytest.mark.gpu

N_RAYS, K_HYP, NS, NI = 35, 6, 64, 128          # 18 + 17 rays: uneven shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from scade_amd.synthetic import synthetic_rays
    g = torch.Generator().manual_seed(21)
    rays = synthetic_rays(N_RAYS, seed=20)
    tgt = torch.rand(N_RAYS, 3, generator=g)
    hyp = torch.rand(K_HYP, N_RAYS, 1, generator=g) * 4.9 + 0.1
    mask = (torch.rand(N_RAYS, generator=g) > 0.3).float()
    draws = dict(t_rand=torch.rand(N_RAYS, NS, generator=g), u_coarse=torch.rand(N_RAYS, NI, generator=g),
                 cached_u=torch.rand(N_RAYS, NI, generator=g))
    u_joint = torch.rand(NI, generator=g)
    return rays, tgt, hyp, mask, draws, u_joint


CASES = [  # name, Trainer kwargs, uses mask
    ("single", dict(allreduce="single"), False),
    ("overlap", dict(allreduce="overlap"), False),
    ("wild_mask", dict(allreduce="single", mask_mode="wild"), True),
    ("joint", dict(allreduce="single", is_joint=True), False),
    ("bf16_overlap", dict(allreduce="overlap", precision="bf16"), False),
    # round 5: the exchange in two pieces WITHOUT leaving the joint backward - the coarse network's piece starts
    # behind its own weight gradient + reduce and runs under the fine network's weight gradient
    ("staged", dict(allreduce="staged"), False),
    ("staged_wild_mask", dict(allreduce="staged", mask_mode="wild"), True),
    ("bf16s8_staged", dict(allreduce="staged", precision="bf16-s8"), False),
]


def _run_case(dev, kw, use_mask, a, b, n_total, two_steps=True):
    """-> (reduced gradient bucket of step 1, parameters after 2 steps, loss term of step 1)"""
    from scade_amd.train import Trainer, make_scade_nets
    rays, tgt, hyp, mask, draws, u_joint = _problem()
    coarse, fine = make_scade_nets(dev, seed=5)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=3, scaleshift_lr=1e-3, **kw)
    d = {k: v[a:b].to(dev) for k, v in draws.items()}
    if kw.get("is_joint"):
        d["cached_u"] = u_joint.to(dev)                       # ONE draw for the whole batch (helpers:498-513)
    args = (rays[a:b].to(dev), tgt[a:b].to(dev), hyp[:, a:b].to(dev))
    m = mask[a:b].to(dev) if use_mask else None
    loss, _ = tr.step(*args, img_i=1, mask=m, n_total=n_total, **d)
    grad = tr.bucket.grad.clone()
    if two_steps:
        tr.step(*args, img_i=2, mask=m, n_total=n_total, **d)
    return grad.cpu(), tr.bucket.data.clone().cpu(), float(loss)


IMG_H, IMG_W = 21, 23          # 483 rays: uneven shards, a ragged last chunk


def _render_image(dev, shard):
    """One full-image test render (run_scade_scannet.py:80-155) through scade_amd.render."""
    import scade_amd as S
    from scade_amd.train import make_scade_nets
    coarse, fine = make_scade_nets(dev, seed=5)
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
    intr = torch.tensor([30.0, 30.0, IMG_W / 2, IMG_H / 2], device=dev)
    c2w = torch.eye(4, device=dev)[:3, :4].contiguous()
    with torch.no_grad():
        rgb, disp, acc, extras = S.render(IMG_H, IMG_W, intr, chunk=128, c2w=c2w, near=0.1, far=5.0, use_viewdirs=True,
                                          shard_group=True if shard else None, network_fn=coarse, network_fine=fine,
                                          network_query_fn=query, N_samples=NS, N_importance=NI, perturb=0.)
    return {"rgb": rgb.cpu(), "disp": disp.cpu(), "acc": acc.cpu(), "depth": extras["depth_map"].cpu(),
            "rgb0": extras["rgb0"].cpu(), "z_std": extras["z_std"].cpu(), "keys": sorted(extras)}


def _run_driver(out_dir, rank):
    """scade_amd.driver.train_scene on the tiny scene the parent wrote: 20 iterations of 96-ray batches."""
    from scade_amd import driver, scene
    data = scene.load_scene_scannet(os.path.join(out_dir, "scene"), "dump", num_hypothesis=4)
    res = driver.train_scene(data, os.path.join(out_dir, f"ckpt_rank{rank}"), "t", "tiny", num_iterations=20, N_rand=97,
                             i_weights=1000, i_print=10, scaleshift_lr=1e-4, test_chunk=256, log=lambda *_: None)
    return {"params": res["trainer"].bucket.data.clone().cpu(), "trace": res["trace"], "test": res["test"]}


def _worker(rank, world, port, out_dir, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = rank if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from scade_amd.parallel import shard_range, shared_uniform
    a, b = shard_range(N_RAYS, rank, world)
    out = {}
    for name, kw, use_mask in CASES:
        out[name] = _run_case(dev, kw, use_mask, a, b, N_RAYS)
    # in-kernel draws (no injection): both ranks step on the SAME rays, built under the SAME torch seed
    from scade_amd.train import Trainer, make_scade_nets
    coarse, fine = make_scade_nets(dev, seed=5)
    tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=3)
    rays_, tgt_, hyp_ = _problem()[:3]
    aux = tr.step(rays_[:16].to(dev), tgt_[:16].to(dev), hyp_[:, :16].to(dev), n_total=32)[1]
    out["own_draws"] = (aux["ret"]["z_vals0"].cpu(), aux["ret"]["u"].cpu())
    torch.manual_seed(100 + rank)
    out["shared_u"] = shared_uniform((NI,), dev).cpu()
    out["render"] = _render_image(dev, shard=True)          # test render sharded over the ranks
    out["driver"] = _run_driver(out_dir, rank)               # the thin driver, rays of every batch sharded
    if backend == "nccl":
        # the whole sharded step as ONE HIP graph (RCCL all-reduce captured), joint exchange included
        from scade_amd.graphs import GraphedTrainer
        from scade_amd.train import Trainer, make_scade_nets
        rays, tgt, hyp, mask, draws, u_joint = _problem()
        for name, kw in (("graph", {}), ("graph_joint", dict(is_joint=True)), ("graph_staged", dict(allreduce="staged"))):
            res = []
            for graphed in (False, True):
                coarse, fine = make_scade_nets(dev, seed=5)
                tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=3, **kw)
                d = {k: v[a:b].to(dev) for k, v in draws.items()}
                if kw.get("is_joint"):
                    d["cached_u"] = u_joint.to(dev).expand(b - a, NI).contiguous()
                args = (rays[a:b].to(dev), tgt[a:b].to(dev), hyp[:, a:b].to(dev))
                if graphed:
                    gt = GraphedTrainer(tr, b - a, K_HYP, inject_draws=True, n_total=N_RAYS)
                    for i in range(3):
                        gt.step(*args, img_i=i, **d)
                else:
                    for i in range(3):
                        tr.step(*args, img_i=i, n_total=N_RAYS, **d)
                res.append(tr.bucket.data.clone().cpu())
            out[name] = res
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_trainer_two_ranks_matches_single_process(dev, tmp_path):
    from test_gpu_driver import write_smooth_scene
    write_smooth_scene(str(tmp_path / "scene"))
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    # every rank must exit cleanly, teardown included.  (Round 2 tolerated ranks that "died in the teardown" of an
    # RCCL group whose collectives had been graph-captured.  Round 3 found the cause - the process group's
    # watchdog thread polling events while a GLOBAL-mode capture was open, scade_amd/graphs.py _capture_mode - and
    # removed it, so an abnormal exit is a failure again.)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert torch.equal(outs[0]["shared_u"], outs[1]["shared_u"]), "sample_pdf_joint's u is one draw for all ranks"
    # per-rank distinct jitter / u streams - the Philox key carries the rank (Trainer.draw_key)
    (z0, u0), (z1, u1) = outs[0]["own_draws"], outs[1]["own_draws"]
    assert float((u0 == u1).float().mean()) < 0.01 and float((z0 == z1).float().mean()) < 0.05, "ranks drew the same stream"
    assert float(u0.min()) >= 0 and float(u0.max()) < 1 and abs(float(u1.mean()) - 0.5) < 0.05
    for name, kw, use_mask in CASES:
        g0, p0, l0 = outs[0][name]
        g1, p1, l1 = outs[1][name]
        assert torch.equal(g0, g1) and torch.equal(p0, p1), f"{name}: ranks diverged"
        g, p, l = _run_case(dev, kw, use_mask, 0, N_RAYS, None)          # the whole batch, one process
        tol = 2e-2 if kw.get("precision") in ("bf16", "bf16-s8") else 2e-5   # 16-bit: tile composition changes roundings
        n_net = g.numel() - 6
        assert rel_l2(g0[:n_net], g[:n_net]) < tol, f"{name}: network gradients {rel_l2(g0[:n_net], g[:n_net]):.2e}"
        assert rel_l2(g0[n_net:], g[n_net:]) < max(tol, 1e-4), f"{name}: scale/shift gradients"
        assert float(g[n_net + 1].abs()) > 0 and float(g[n_net].abs()) == 0, "step 1 touches image 1 only"
        assert abs(l0 + l1 - l) < max(tol, 1e-5) * abs(l), f"{name}: rank loss terms {l0}+{l1} vs {l}"
        if kw.get("precision") not in ("bf16", "bf16-s8"):
            # two Adam steps from identical states: the first update is lr * sign(g), so every gradient
            # element that is summation-order noise around zero moves its weight by +-lr in either run
            # (2 lr = 1e-3 against weights of ~0.1); the gradient comparison above is the sharp test
            assert rel_l2(p0, p) < 1e-3, f"{name}: parameters after two steps {rel_l2(p0, p):.2e}"
    # the staged exchange is the single-bucket exchange in another launch order
    assert rel_l2(outs[0]["staged"][0], outs[0]["single"][0]) < 2e-5 and rel_l2(outs[0]["staged"][1], outs[0]["single"][1]) < 1e-3
    assert rel_l2(outs[0]["staged_wild_mask"][0], outs[0]["wild_mask"][0]) < 2e-5
    # sharded test render: every rank holds the whole image, bit-identical to the one-process render
    want = _render_image(dev, shard=False)
    for r in range(world):
        got = outs[r]["render"]
        assert got["rgb"].shape == (IMG_H, IMG_W, 3)
        for k in ("rgb", "disp", "acc", "depth", "rgb0", "z_std"):
            assert torch.equal(torch.nan_to_num(got[k]), torch.nan_to_num(want[k])), f"rank {r}: sharded render {k} differs"
        assert "z_vals" not in got["keys"] and "z_vals" in want["keys"], "only the per-pixel maps travel by default"
    # the driver: 97-ray batches split 49 + 48, same image / pixel stream on both ranks -> identical parameters
    # on both, the one-process run's parameters up to summation order, the same test metrics everywhere
    d0, d1 = outs[0]["driver"], outs[1]["driver"]
    assert torch.equal(d0["params"], d1["params"]), "driver: ranks diverged"
    assert d0["test"] == d1["test"], "every rank holds the whole test image -> the same metrics"
    from scade_amd import driver, scene
    data = scene.load_scene_scannet(str(tmp_path / "scene"), "dump", num_hypothesis=4)
    one = driver.train_scene(data, str(tmp_path / "ckpt_one"), "t", "tiny", num_iterations=20, N_rand=97, i_weights=1000,
                             i_print=10, scaleshift_lr=1e-4, test_chunk=256, log=lambda *_: None)
    assert rel_l2(d0["params"], one["trainer"].bucket.data.cpu()) < 2e-2, "driver: sharded vs one process after 20 Adam steps"
    assert abs(d0["test"]["psnr"] - one["test"]["psnr"]) < 1.0
    if backend == "nccl":
        for name in ("graph", "graph_joint", "graph_staged"):
            eager, graphed = outs[0][name]
            assert rel_l2(graphed, eager) < 1e-5, f"{name}: graphed sharded step diverges from eager"
            assert torch.equal(outs[0][name][1], outs[1][name][1])
