"""CPU, world_size 2 / 4 / 8, gloo: the ray-sharded train step (scade_amd/parallel.py) reproduces
the single-process gradients.  The HIP kernels cannot run here, so the per-rank compute is
the CPU oracle; what is under test is the host logic: sharding, loss normalisation, the
flat gradient bucket and its all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scade_oracle as O
from scade_amd.parallel import (FlatParams, batch_share, gather_rows, render_rays_sharded, seed_rank_streams,
                                shard_batch, shard_range, staircase_lr)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_staircase_lr():
    # the argument is the reference's loop index i (counts from 1; Trainer passes it + 1)
    assert staircase_lr(5e-4, 0.1, 400000, 1) == 5e-4
    assert staircase_lr(5e-4, 0.1, 400000, 399999) == 5e-4
    assert abs(staircase_lr(5e-4, 0.1, 400000, 400000) - 5e-5) < 1e-12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(N=16):
    K = 5
    rays = O.synthetic_rays(N, seed=3)
    g = torch.Generator().manual_seed(4)
    tgt = torch.rand(N, 3, generator=g)
    hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
    t_rand = torch.rand(N, 16, generator=g)
    u = torch.rand(N, 24, generator=g)
    return rays, tgt, hyp, t_rand, u


def _loss(pc, pf, scale, shift, rays, tgt, hyp, t_rand, u):
    ret = O.render_rays(rays, pc, pf, torch.zeros(3), torch.tensor(0.2), n_samples=16, n_importance=24,
                        t_rand=t_rand, u_coarse=u, u_fine=u)
    return O.train_loss(ret, tgt, hyp * scale + shift)[0]


def _params():
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    tensors = [v.clone().requires_grad_(True) for v in pc.values()] + \
              [v.clone().requires_grad_(True) for v in pf.values()] + \
              [torch.ones(1, requires_grad=True), torch.zeros(1, requires_grad=True)]
    n = len(pc)
    return tensors, dict(zip(pc.keys(), tensors[:n])), dict(zip(pf.keys(), tensors[n:2 * n])), tensors[-2], tensors[-1]


def _worker(rank, world, port, out_dir, n_rays, pieces):
    """One rank of the Trainer's exchange: ONE bucket [coarse | fine | scale | shift], every loss
    term weighted by n_local / N_total, sum-all-reduce (whole, or in two async pieces), no 1/world."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    rays, tgt, hyp, t_rand, u = _problem(n_rays)
    tensors, pc, pf, scale, shift = _params()
    flat = FlatParams(tensors)                                        # the networks AND scale/shift
    flat.broadcast_params(0)
    flat.zero_grad()
    a, b = shard_range(rays.shape[0], rank, world)
    r, t, h = shard_batch(rays, tgt, hyp, rank, world)
    share = batch_share(b - a, n_rays)
    assert abs(share - (b - a) / n_rays) < 1e-12
    loss = _loss(pc, pf, scale, shift, r, t, h, t_rand[a:b], u[a:b]) * share
    loss.backward()
    assert tensors[0].grad.data_ptr() == flat.grad.data_ptr()       # accumulated in place
    if pieces == "staged":
        # Trainer(allreduce="staged"): the host logic of the two-piece exchange that keeps the joint backward - the
        # backward's hook sends a network's piece when its gradient is complete (_send_net_grads: the last network's
        # piece takes the scale / shift rows along), reduce_grads sends what the hook has not and waits for all
        import types
        from scade_amd.train import Trainer
        n_c = sum(v.numel() for v in pc.values())
        net_c, net_f = object(), object()
        flat._sinks = [(net_c, 0), (net_f, n_c)]
        stub = types.SimpleNamespace(bucket=flat, n_net=2 * n_c, force_allreduce=False, allreduce="staged", sharded=True,
                                     coarse_stream=None, _staged_works=[], _staged_done=[])
        Trainer._send_net_grads(stub, net_c)
        assert stub._staged_done == [(0, n_c)] and len(stub._staged_works) == 1
        Trainer._send_net_grads(stub, net_f)
        assert stub._staged_done[1] == (n_c, flat.numel - n_c), "the fine piece takes the scale / shift rows along"
        Trainer.reduce_grads(stub)
        assert stub._staged_works == [] and stub._staged_done == []
    elif pieces == "staged_late":
        # the same mode when the backward was NOT a joint one (no hook fired for the fine network): reduce_grads
        # sends the complement of what went out
        import types
        from scade_amd.train import Trainer
        n_c = sum(v.numel() for v in pc.values())
        net_c = object()
        flat._sinks = [(net_c, 0)]
        stub = types.SimpleNamespace(bucket=flat, n_net=2 * n_c, force_allreduce=False, allreduce="staged", sharded=True,
                                     coarse_stream=None, _staged_works=[], _staged_done=[])
        Trainer._send_net_grads(stub, net_c)
        Trainer.reduce_grads(stub)
    elif pieces:
        n_c = sum(v.numel() for v in pc.values())
        works = flat.allreduce_grads_async([(0, n_c, None), (n_c, flat.numel - n_c, None)])
        assert len(works) == 2
        flat.wait_all(works)
    else:
        assert flat.allreduce_grads() == 1.0 / world                  # legacy factor, unused here
    # the Philox key of the Trainer's in-kernel draws (host logic): every rank builds its Trainer under the SAME
    # torch seed (identical weight init), so the key must carry the rank; it is derived at first use, so a later
    # seed_rank_streams / manual_seed still selects the stream
    import types
    from scade_amd.train import Trainer
    torch.manual_seed(7)
    keys = []
    for reseed in (None, 8):
        stub = types.SimpleNamespace(_draw_key=None)
        if reseed is not None:
            torch.manual_seed(reseed)
        keys.append(Trainer.draw_key(stub))
        assert Trainer.draw_key(stub) == keys[-1], "stable once derived"
    torch.save((flat.grad.clone(), loss.detach(), keys), os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays,pieces", [
    (2, 16, False), (2, 17, False), (2, 17, True), (2, 17, "staged"), (2, 16, "staged_late"),
    # the rank counts of BASELINE configs[3] / [4] (no box of the pool has more than one GPU: the logic above two
    # ranks runs here) - uneven shards: 35 rays = 9 + 9 + 9 + 8, 43 rays = 6 + 6 + 6 + 5 + 5 + 5 + 5 + 5
    (4, 35, False), (4, 35, "staged"), (8, 43, False), (8, 43, "staged"), (8, 40, "staged_late")])
def test_sharded_gradients_match_single_process(tmp_path, world, n_rays, pieces):
    """Even and uneven ray shards over 2 / 4 / 8 ranks: the summed share-weighted gradients are the single-process ones."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n_rays, pieces), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(world)]
    g0 = outs[0][0]
    keys = {k for _, _, ks in outs for k in ks}
    assert len(keys) == 2 * world, "in-kernel draw keys: distinct per rank AND per torch seed at first use"
    for g, _, _ in outs[1:]:
        assert torch.equal(g0, g), "all ranks must hold identical reduced gradients"
    rays, tgt, hyp, t_rand, u = _problem(n_rays)
    tensors, pc, pf, scale, shift = _params()
    want = _loss(pc, pf, scale, shift, rays, tgt, hyp, t_rand, u)
    want.backward()
    ref = torch.cat([t.grad.reshape(-1) if t.grad is not None else torch.zeros(t.numel()) for t in tensors])
    err = (g0 - ref).norm() / ref.norm()
    assert err < 1e-5, f"sharded vs single-process gradient rel-L2 {err:.3e}"
    assert float(ref[-2:].abs().min()) > 0, "scale / shift gradients ride in the same bucket"
    assert abs(sum(float(l) for _, l, _ in outs) - float(want)) < 1e-5 * abs(float(want)), "rank terms sum to the global loss"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shards_of_a_1000_ray_batch(world):
    """The slicing every rank applies to a global batch (rays, targets, hypothesis-major [K,N,1] depth hypotheses,
    wild mask): the shards tile the batch, the hypothesis slices are the matching COLUMNS of every hypothesis row, and
    the n_local / N_total shares sum to one - at 1000 rays (125 per rank at world 8) and at 1001 (uneven)."""
    for n in (1000, 1001):
        g = torch.Generator().manual_seed(n)
        rays, tgt = torch.rand(n, 11, generator=g), torch.rand(n, 3, generator=g)
        hyp, mask = torch.rand(40, n, 1, generator=g), (torch.rand(n, generator=g) > 0.5).float()
        parts = [shard_batch(rays, tgt, hyp, r, world, mask=mask) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts], 0), rays)
        assert torch.equal(torch.cat([p[1] for p in parts], 0), tgt)
        assert torch.equal(torch.cat([p[2] for p in parts], 1), hyp), "dim 1 of [K,N,1] is the ray axis"
        assert torch.equal(torch.cat([p[3] for p in parts], 0), mask)
        sizes = [p[0].shape[0] for p in parts]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
        if n == 1000 and world == 8:
            assert sizes == [125] * 8
        assert all(p[2].shape == (40, s, 1) for p, s in zip(parts, sizes))
        assert abs(sum(s / n for s in sizes) - 1.0) < 1e-12


def test_flat_segments_share_the_bucket():
    ps = [torch.nn.Parameter(torch.randn(6)), torch.nn.Parameter(torch.randn(2, 1)), torch.nn.Parameter(torch.randn(2, 1))]
    flat = FlatParams(ps)
    nets, ss = flat.segment(0, 6), flat.segment(6, 4)
    (ps[0].sum() + 3 * ps[1].sum() + 5 * ps[2].sum()).backward()
    assert torch.equal(nets.grad, torch.ones(6)) and torch.equal(ss.grad, torch.tensor([3., 3., 5., 5.]))
    ss.data.zero_()
    assert float(ps[1].abs().sum() + ps[2].abs().sum()) == 0.0 and nets.data.data_ptr() == flat.data.data_ptr()
    ss.zero_grad()
    assert float(flat.grad[6:].abs().sum()) == 0.0 and float(flat.grad[:6].sum()) == 6.0
    with pytest.raises(ValueError):
        flat.segment(8, 4)


def test_flatparams_views_and_zero_grad():
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    before = [p.detach().clone() for p in ps]
    flat = FlatParams(ps)
    assert flat.numel == 17
    for p, b in zip(ps, before):
        assert torch.equal(p.detach(), b)
    (ps[0].sum() * 2 + ps[1].sum()).backward()
    assert torch.equal(flat.grad, torch.cat([torch.full((12,), 2.0), torch.ones(5)]))
    ps[0].grad = None
    flat.zero_grad()
    assert ps[0].grad is not None and float(flat.grad.abs().sum()) == 0.0
    flat.data.mul_(0)
    assert float(ps[1].abs().sum()) == 0.0


# ---------------------------------------------------------------- is_joint exchange
class _JointShardedCpu(torch.autograd.Function):
    """CPU stand-in with the SAME structure as scade_amd.ops.CarveJointShardedFn (kernel phases
    replaced by torch ops): column means of the shard -> parallel.combine_shard_means -> min over
    K / mean over samples; backward = this shard's part of the global gradient."""

    @staticmethod
    def forward(ctx, pred, hyp):
        from scade_amd.parallel import combine_shard_means
        d = (pred[None] - hyp).abs()                       # [K,n,P]
        means = d.mean(dim=1)                              # [K,P] over this shard's rays
        share, world = combine_shard_means(means, pred.shape[0], n_total=_joint_problem()[0].shape[0])
        best, arg = means.min(dim=0)
        ctx.save_for_backward(pred, hyp, arg)
        ctx.factor = share
        return best.mean()

    @staticmethod
    def backward(ctx, g):
        pred, hyp, arg = ctx.saved_tensors
        n, P = pred.shape
        sel = hyp[arg, :, 0].T if hyp.shape[-1] == 1 else None      # [n,P] hypothesis of the argmin
        sgn = torch.sign(pred - sel)
        gp = sgn * (g * ctx.factor / (n * P))
        gh = torch.zeros_like(hyp)
        gh[:, :, 0].index_put_((arg[None, :].expand(n, P).reshape(-1),
                                torch.arange(n)[:, None].expand(n, P).reshape(-1)), -gp.reshape(-1),
                               accumulate=True)
        return gp, gh


def _joint_problem():
    g = torch.Generator().manual_seed(11)
    N, K, P = 17, 6, 12                                    # uneven shards: 9 + 8 rays (world 2) .. 3 + 2 x 7 (world 8)
    pred = (torch.rand(N, P, generator=g) * 4 + 0.5)
    hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
    return pred, hyp


def _joint_worker(rank, world, port, out_dir):
    from scade_amd.parallel import shared_uniform
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pred, hyp = _joint_problem()
    a, b = shard_range(pred.shape[0], rank, world)
    p = pred[a:b].clone().requires_grad_(True)
    h = hyp[:, a:b].clone().requires_grad_(True)
    loss = _JointShardedCpu.apply(p, h)
    loss.backward()
    torch.manual_seed(100 + rank)                          # different local streams ...
    u = shared_uniform((8,), torch.device("cpu"))          # ... one shared draw
    torch.save((loss.detach(), p.grad, h.grad, u), os.path.join(out_dir, f"j{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_joint_space_carving_exchange_matches_single_process(tmp_path, world):
    mp.spawn(_joint_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"j{r}.pt")) for r in range(world)]
    pred, hyp = _joint_problem()
    p = pred.clone().requires_grad_(True)
    h = hyp.clone().requires_grad_(True)
    want = O.compute_space_carving_loss(p, h, is_joint=True)
    want.backward()
    for r in range(world):
        assert abs(float(outs[r][0]) - float(want)) < 1e-6 * abs(float(want)), "every rank holds the GLOBAL loss"
    # the trainer SUMS the ranks' gradients: here pred/hyp shards are disjoint inputs, so the
    # per-shard gradient must equal the matching slice of the full gradient
    gp = torch.cat([outs[r][1] for r in range(world)], 0)
    gh = torch.cat([outs[r][2] for r in range(world)], 1)
    assert torch.allclose(gp, p.grad, rtol=1e-5, atol=1e-8)
    assert torch.allclose(gh, h.grad, rtol=1e-5, atol=1e-8)
    for r in range(1, world):
        assert torch.equal(outs[0][3], outs[r][3]), "sample_pdf_joint's u must be one draw for all ranks"


def _fake_render(rows):
    """Stands in for batchify_rays on CPU: per-ray maps that depend on the ray row only."""
    return {"rgb_map": rows[:, :3] * 2.0, "depth_map": rows.sum(-1), "z_vals": rows[:, :4].repeat(1, 2)}


_RENDER_RAYS = 17


def _render_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays = O.synthetic_rays(_RENDER_RAYS, seed=9)             # 17 rays: shards of 9 + 8 (world 2) .. 3 + 7 x 2 (world 8)
    a, b = shard_range(_RENDER_RAYS, rank, world)
    full = gather_rows(rays[a:b].contiguous(), _RENDER_RAYS)
    img = render_rays_sharded(rays, _fake_render)             # default keys: per-pixel maps only
    everything = render_rays_sharded(rays, _fake_render, keys=None)
    seed_rank_streams(5)
    # by value (numpy), not as shared-memory tensors: a tensor travels as a file descriptor that the parent must
    # fetch from this process while it is still alive - a worker that exits first resets the connection
    npd = lambda d: {k: v.numpy() for k, v in d.items()}
    q.put((rank, full.numpy(), npd(img), npd(everything), torch.rand(4).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_test_render_gathers_whole_image_on_every_rank(world):
    """the test render shards the H*W rays over the ranks and gathers the image."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_render_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    rays = O.synthetic_rays(_RENDER_RAYS, seed=9)
    want = _fake_render(rays)
    draws = {}
    for rank, full, img, everything, draw in res:
        full, draw = torch.from_numpy(full), torch.from_numpy(draw)
        img = {k: torch.from_numpy(v) for k, v in img.items()}
        everything = {k: torch.from_numpy(v) for k, v in everything.items()}
        draws[rank] = draw
        assert torch.equal(draw, torch.rand(4, generator=torch.Generator().manual_seed(5 + rank))), "seed + rank"
        assert torch.equal(full, rays), f"rank {rank}: gather_rows"
        assert sorted(img) == ["depth_map", "rgb_map"], sorted(img)
        assert sorted(everything) == ["depth_map", "rgb_map", "z_vals"]
        for k in everything:
            assert torch.equal(everything[k], want[k]), (rank, k)
    assert len({tuple(d.tolist()) for d in draws.values()}) == world, "per-rank distinct jitter streams (SURVEY 8e)"
    # one process (no group): the function is the identity wrapper
    assert torch.equal(render_rays_sharded(rays, _fake_render)["z_vals"], want["z_vals"])
