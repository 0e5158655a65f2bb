"""GPU: the opt-in split-precision inference path (NeRF.inference_precision = "f16x3":
x ~= h + l*2^-11 in fp16, three f16 MFMAs per product, fp32 accumulate) is held to the SAME
parity bar as the exact fp32 kernels."""
import pytest
import torch

import scade_amd as S
from conftest import assert_close, load_golden, rel_l2
from oracle import scade_oracle as O
from test_oracle_golden import f2_params, f6_params
from test_gpu_ops import make_net
from test_gpu_render import build, check_ret, stagewise

pytestmark = pytest.mark.gpu


def fast(net):
    net.inference_precision = "f16x3"
    return net


def test_f16x3_forward_golden(dev):
    g = load_golden("f2_mlp")
    net = fast(make_net(f2_params(g), dev))
    with torch.no_grad():
        out = net(g["x"].to(dev))
    assert_close(out, g["out"], rtol=1e-4, atol=1e-5, what="NeRF.forward f16x3")
    assert rel_l2(out, g["out"]) < 2e-6
    # with grad enabled the module silently uses the exact training kernels
    out2 = net(g["x"].to(dev))
    assert out2.requires_grad
    assert_close(out2, g["out"], rtol=1e-4, atol=1e-5, what="training forward stays exact")


def test_f16x3_points_ragged_and_edge_sizes(dev):
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = fast(make_net(params, dev))
    torch.manual_seed(3)
    N, Sm = 37, 5
    pts = torch.rand(N, Sm, 3) * 6 - 3
    vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    bbc, bbs = torch.tensor([0.1, -0.2, 0.3]), torch.tensor(0.2)
    want = O.run_network(pts, vd, lambda e: O.nerf_forward(params, e), bbc, bbs)
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    with torch.no_grad():
        got = S.run_network(pts.to(dev), vd.to(dev), torch.empty(0, device=dev), net, e, ed, bbc.to(dev),
                            bbs.to(dev))
        for P in (0, 1, 63, 64, 65, 129):
            o = net(g["x"][:P].to(dev))
            assert o.shape == (P, 4)
            if P:
                assert_close(o, g["out"][:P], rtol=1e-4, atol=1e-5, what=f"f16x3 P={P}")
    assert_close(got, want, rtol=1e-4, atol=1e-5, what="run_network f16x3")


def test_f16x3_large_activations_and_small_values(dev):
    """Dynamic range: activations of a few thousand and tiny weights stay accurate (the low
    plane is pre-scaled by 2^11, so it never falls into the fp16 subnormal range)."""
    params = O.nerf_init(7)
    params["pts_linears.0.weight"] = params["pts_linears.0.weight"] * 300.0      # |h0| ~ 1e3
    params["pts_linears.1.weight"] = params["pts_linears.1.weight"] * 1e-3
    params["pts_linears.2.weight"] = params["pts_linears.2.weight"] * 30.0
    net = fast(make_net(params, dev))
    torch.manual_seed(4)
    x = torch.cat([O.embed(torch.rand(200, 3) * 2 - 1, 9),
                   torch.nn.functional.normalize(torch.randn(200, 3), dim=-1)], -1)
    want = O.nerf_forward({k: v.double() for k, v in params.items()}, x.double())
    with torch.no_grad():
        got = net(x.to(dev))
        net.inference_precision = "f32"
        exact = net(x.to(dev))
    e_fast, e_exact = rel_l2(got, want), rel_l2(exact, want)
    assert e_fast < 3e-6 and e_fast < 4 * e_exact + 1e-7, (e_fast, e_exact)


def test_f16x3_render_rays_golden(dev):
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    fast(coarse); fast(fine)
    with torch.no_grad():
        ret = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, embedded_cam=torch.empty(0, device=dev),
                            N_importance=128, network_fine=fine, perturb=0., retraw=True)
        want = {k[4:]: v for k, v in g.items() if k.startswith("det/")}
        check_ret(ret, want, "det f16x3")
        stagewise(dev, want, g["rays"], fine, query, want["u"])
        ret2 = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                             perturb=0., retraw=True)
    for k in ret:
        assert torch.equal(torch.nan_to_num(ret[k]), torch.nan_to_num(ret2[k])), f"non-deterministic {k}"


def test_f16x3_repack_and_bad_mode(dev):
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = fast(make_net(params, dev))
    x = g["x"].to(dev)
    with torch.no_grad():
        a = net(x).clone()
        net.pts_linears[2].weight.mul_(0.5)
        b = net(x)
    params["pts_linears.2.weight"] = params["pts_linears.2.weight"] * 0.5
    assert_close(b, O.nerf_forward(params, g["x"]), rtol=1e-4, atol=1e-5, what="f16x3 after update")
    assert not torch.allclose(a, b)
    net.inference_precision = "fp8"
    with pytest.raises(ValueError):
        with torch.no_grad():
            net(x)


# ---------------------------------------------------------------- split-precision TRAINING mode
def sub(g):
    f = g.flatten()
    return f if f.numel() <= 4096 else f[::97]


def rows24_to_fp32(acts, P):
    """The f16x3 forward's saved rows (round 5: fp16 h plane [P,256] + e5m2 l plane [P,256] inside each row slot,
    x = h + l * 2^-11; include/scade_hip.h, scade_mlp_fwd_f16 mode + 2) as the fp32 rows the exact backward reads:
    the ten row slots unpacked, everything behind them (embedding rows, alpha_pre, ReLU sign words) as it is."""
    out = acts.clone()
    raw = acts.view(torch.uint8)
    for s in range(10):
        b = raw[s * P * 1024:(s + 1) * P * 1024]
        h = b[:P * 512].view(torch.float16).view(P, 256).float()
        l = b[P * 512:P * 768].view(torch.float8_e5m2).view(P, 256).float()
        out[s * P * 256:(s + 1) * P * 256] = (h + l / 2048.0).reshape(-1)
    return out


def test_f16x3_training_gradients_golden(dev):
    """train_precision='f16x3': forward + dgrad on f16 MFMAs (per-point gradient scale), exact
    wgrad -- same gradient bar as the exact backward (tests/test_gpu_train.py)."""
    from test_gpu_train import grad_close
    g = load_golden("f2_mlp")
    net = make_net(f2_params(g), dev)
    net.train_precision = "f16x3"
    out = net(g["x"].to(dev))
    # tiny upstream gradients (mean-loss scale) must survive the fp16 dgrad planes
    (out * g["G"].to(dev) * 1e-6).sum().backward()
    assert_close(out, g["out"], rtol=1e-4, atol=1e-5, what="f16x3 training forward")
    for k, p in net.named_parameters():
        grad_close(sub(p.grad) * 1e6, g["grad/" + k], f"f16x3 d/d{k}")
        assert rel_l2(sub(p.grad) * 1e6, g["grad/" + k]) < 2e-5, k


def test_f16x3_training_many_chunks_and_zero_rows(dev):
    from test_gpu_train import grad_close
    params = O.nerf_init(5)
    net = make_net(params, dev)
    net.train_precision = "f16x3"
    torch.manual_seed(9)
    P = 4000
    pts = torch.rand(P, 3) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)
    x = torch.cat([O.embed(pts, 9), vd], -1)
    G = torch.randn(P, 4) * torch.logspace(-9, 0, P)[:, None]      # 9 decades of per-point scale
    G[::7] = 0.0                                                    # points without gradient
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    (O.nerf_forward(po, x) * G).sum().backward()
    out = net(x.to(dev))
    (out * G.to(dev)).sum().backward()
    for k, p in net.named_parameters():
        grad_close(p.grad, po[k].grad, f"d/d{k}", rtol=2e-4, scale_atol=5e-5)
        assert rel_l2(p.grad, po[k].grad) < 2e-5, k


@pytest.mark.parametrize("P", [1, 15, 17, 33, 47, 65, 130, 1000])
def test_f16x3_training_gradients_ragged_and_odd_stage_counts(dev, P):
    """The split-precision backward at sizes that exercise its edges (round 5): a lone partial stage, an odd number of
    16-point stages (the ring runs in pairs: the partner is range-check zeros), ragged 64-point dgrad tiles whose row
    copies ride in the next layer's k-loop, tile-major LDS images cut by the descriptor's range check - against autograd
    of the oracle at the bars of the large-P tests."""
    from test_gpu_train import grad_close
    params = O.nerf_init(6)
    net = make_net(params, dev)
    net.train_precision = "f16x3"
    g = torch.Generator().manual_seed(100 + P)
    pts = torch.rand(P, 3, generator=g) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    x = torch.cat([O.embed(pts, 9), vd], -1)
    G = torch.randn(P, 4, generator=g) * 1e-3
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    (O.nerf_forward(po, x) * G).sum().backward()
    out = net(x.to(dev))
    (out * G.to(dev)).sum().backward()
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        # (per element against the tensor's scale: with a handful of points an element is a short signed sum)
        grad_close(p.grad, po[k].grad, f"P={P} d/d{k}", rtol=2e-4, scale_atol=2e-4)
        assert rel_l2(p.grad, po[k].grad) < 3e-5, (k, rel_l2(p.grad, po[k].grad))


def test_f16x3_train_step_golden(dev):
    """Full train step in split precision.  The loss is held to the golden value.  The gradients
    are checked in two ways: (i) STRICT - against the exact fp32 backward kernels evaluated on the
    very workspace (activations, ReLU sign words, upstream gradient) of this step; (ii) against the
    oracle's golden gradients norm-wise: a ReLU network's gradient is discontinuous where a
    pre-activation crosses zero, and two correct forwards that differ by 1e-7 disagree on the sign
    of a handful of units (2 of 4.2 M sign bits on this batch), each of which moves the gradient of
    a 32-ray batch by up to a percent."""
    from scade_amd import ops
    from test_gpu_train import train_step
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    coarse.train_precision = fine.train_precision = "f16x3"
    scale = torch.ones(1, device=dev, requires_grad=True)
    shift = torch.zeros(1, device=dev, requires_grad=True)
    cap = {}
    orig = ops.mlp_bwd_f16

    def spy(packed, packed_t_f16, acts, g_out, wgrad_f16=True, out=None):
        flat = orig(packed, packed_t_f16, acts, g_out, wgrad_f16, out=out)
        cap[g_out.numel() // 4] = (acts, g_out, flat)
        return flat

    ops.mlp_bwd_f16 = spy
    try:
        ret, loss = train_step(dev, g, coarse, fine, query, scale, shift)
        loss.backward()
    finally:
        ops.mlp_bwd_f16 = orig
    assert_close(loss, g["train/loss"], rtol=1e-4, atol=1e-7, what="loss")
    for P, net in ((32 * 64, coarse), (32 * 192, fine)):
        acts, g_out, flat = cap[P]
        exact = ops.mlp_bwd(net.packed(), net.packed_t(), rows24_to_fp32(acts, P), g_out)
        assert rel_l2(flat, exact) < 2e-5, (P, rel_l2(flat, exact))
        got = torch.cat([p.grad.reshape(-1) for p in net.ordered_params()])
        assert torch.equal(got, flat)
    for k, p in coarse.named_parameters():
        want = g[f"grad_coarse/{k}"]
        got = sub(p.grad) if p.grad is not None else torch.zeros_like(want)
        if float(want.abs().max()) == 0.0:
            assert float(got.abs().max()) == 0.0
        else:
            assert rel_l2(got, want) < 3e-2, (k, rel_l2(got, want))
