"""GPU: the opt-in single-plane 16-bit inference path (NeRF.inference_precision = "f16" / "bf16":
one 16-bit operand per value, one MFMA per product, fp32 accumulate - BASELINE.json config 5's
"bf16 MFMA path").  This is ordinary mixed precision, so it is NOT held to the 1e-4 parity bar
of the exact kernels: the bounds are a relative-L2 error against the oracle and the north-star
acceptance criterion "PSNR within 0.05 dB of the reference render"."""
import pytest
import torch

import scade_amd as S
from conftest import load_golden, rel_l2
from oracle import scade_oracle as O
from test_oracle_golden import f2_params, f6_params
from test_gpu_ops import make_net
from test_gpu_render import build

pytestmark = pytest.mark.gpu

# relative-L2 bound of one MLP evaluation: ~12 chained layers of 2^-11 (fp16) / 2^-8 (bf16) operand rounding
BOUND = {"f16": 2e-3, "bf16": 1.5e-2}


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_forward_golden(dev, prec):
    g = load_golden("f2_mlp")
    net = make_net(f2_params(g), dev)
    net.inference_precision = prec
    with torch.no_grad():
        out = net(g["x"].to(dev))
    err = rel_l2(out, g["out"])
    assert 1e-6 < err < BOUND[prec], err          # > 1e-6: it really is the 16-bit kernel
    # with grad enabled the module uses the exact training kernels
    out2 = net(g["x"].to(dev))
    assert out2.requires_grad and rel_l2(out2, g["out"]) < 2e-6


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_ragged_sizes_and_row_independence(dev, prec):
    """Any P (the tile is 128 points): rows do not depend on the launch size or on their neighbours."""
    g = load_golden("f2_mlp")
    net = make_net(f2_params(g), dev)
    net.inference_precision = prec
    x = g["x"].to(dev)
    with torch.no_grad():
        full = net(x)
        for P in (0, 1, 127, 128, 129, 255):
            o = net(x[:P])
            assert o.shape == (P, 4)
            assert torch.equal(o, full[:P]), P
        again = net(x)
    assert torch.equal(full, again)
    assert torch.isfinite(full).all()


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_point_tilings_agree(dev, prec):
    """The 16-bit forward / dgrad run 128-point workgroups, or 64-point ones when the launch has fewer
    than one workgroup per CU (lp_pick_point_tiles: P < 32768 on a 256-CU part).  The golden fixtures
    pin the small launches; per-point outputs of a large launch are bit-identical to them (same k
    order, same roundings), and the training gradients agree to summation order."""
    net = make_net(O.nerf_init(4), dev)
    net.inference_precision = prec
    g = torch.Generator().manual_seed(13)
    P, Ps = 40000, 3000
    x = torch.cat([O.embed(torch.rand(P, 3, generator=g) * 2 - 1, 9),
                   torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)], -1).to(dev)
    with torch.no_grad():
        big, small = net(x), net(x[:Ps])
    assert torch.equal(big[:Ps], small)
    net.train_precision = prec
    G = torch.zeros(P, 4)
    G[:Ps] = torch.randn(Ps, 4, generator=g) * 1e-3
    G = G.to(dev)
    grads = []
    for xs, gs in ((x, G), (x[:Ps], G[:Ps])):
        for p in net.parameters():
            p.grad = None
        out = net(xs)
        out.backward(gs)
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
    # same rounded activations and dZ rows in both launches; only the chunking of the point sum differs
    assert rel_l2(grads[0], grads[1]) < 1e-4


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_points_mode_matches_embedded_mode(dev, prec):
    params = O.nerf_init(3)
    net = make_net(params, dev)
    net.inference_precision = prec
    torch.manual_seed(2)
    N, Sm = 45, 7
    pts = torch.rand(N, Sm, 3) * 6 - 3
    vd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    bbc, bbs = torch.tensor([0.1, -0.2, 0.3]), torch.tensor(0.2)
    want = O.run_network(pts, vd, lambda e: O.nerf_forward(params, e), bbc, bbs)
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    with torch.no_grad():
        fused = S.run_network(pts.to(dev), vd.to(dev), torch.empty(0, device=dev), net, e, ed, bbc.to(dev),
                              bbs.to(dev))
        x = torch.cat([O.embed((pts.reshape(-1, 3) - bbc) * bbs, 9),
                       vd[:, None, :].expand(N, Sm, 3).reshape(-1, 3)], -1)
        emb = net(x.to(dev)).reshape(N, Sm, 4)
    assert rel_l2(fused, want) < BOUND[prec]
    assert rel_l2(emb, want) < BOUND[prec]
    # the fused mode evaluates sin/cos with the hardware instructions on an exactly reduced argument,
    # the embedded mode receives the oracle's values: different roundings of the 57 inputs
    assert rel_l2(fused, emb) < BOUND[prec] / 2


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_render_psnr_within_0p05_db(dev, prec):
    """North-star acceptance: PSNR of the render against a ground-truth image within 0.05 dB of
    the reference render's.  Ground truth = exact fp32 render (itself pinned to the oracle by
    tests/test_gpu_render.py) + noise at ~25 dB, a typical ScanNet test PSNR; 8192 rays so that
    the estimate is not dominated by the error-noise cross term."""
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    rays = torch.cat([g["rays"], O.synthetic_rays(8160, seed=11)], 0).to(dev)
    kw = dict(N_importance=128, network_fine=fine, perturb=0.)
    with torch.no_grad():
        exact = S.render_rays(rays, True, coarse, query, 64, **kw)
        coarse.inference_precision = fine.inference_precision = prec
        lp = S.render_rays(rays, True, coarse, query, 64, **kw)
    ref_rgb = exact["rgb_map"].cpu()
    assert rel_l2(ref_rgb[:32], g["det/rgb_map"]) < 1e-4
    torch.manual_seed(0)
    target = ref_rgb + 0.056 * torch.randn_like(ref_rgb)          # ~25 dB
    p_ref, p_lp = psnr(ref_rgb, target), psnr(lp["rgb_map"].cpu(), target)
    assert 20 < p_ref < 30
    print(f"PSNR vs 25 dB target: exact {p_ref:.4f} dB, {prec} {p_lp:.4f} dB")
    assert abs(p_lp - p_ref) < 0.05, (p_ref, p_lp)
    # and the render itself stays close to the exact one (coarse maps see identical sample positions)
    assert psnr(lp["rgb0"].cpu(), exact["rgb0"].cpu()) > (50 if prec == "f16" else 35)
    assert rel_l2(lp["depth0"], exact["depth0"]) < BOUND[prec]
    assert psnr(lp["rgb_map"].cpu(), ref_rgb) > (45 if prec == "f16" else 30)
    for k in ("rgb_map", "depth_map", "acc_map", "pred_hyp", "weights"):
        assert torch.isfinite(lp[k]).all(), k
    assert (lp["z_vals"][:, 1:] >= lp["z_vals"][:, :-1]).all()


def test_lp_repack_after_update_and_bad_mode(dev):
    g = load_golden("f2_mlp")
    params = f2_params(g)
    net = make_net(params, dev)
    net.inference_precision = "f16"
    x = g["x"].to(dev)
    with torch.no_grad():
        a = net(x).clone()
        net.pts_linears[2].weight.mul_(0.5)
        b = net(x)
        net.inference_precision = "bf16"           # switching format re-packs too
        c = net(x)
    params["pts_linears.2.weight"] = params["pts_linears.2.weight"] * 0.5
    want = O.nerf_forward(params, g["x"])
    assert rel_l2(b, want) < BOUND["f16"] and rel_l2(c, want) < BOUND["bf16"]
    assert not torch.allclose(a, b) and not torch.equal(b, c)
    net.inference_precision = "int8"
    with pytest.raises(ValueError):
        with torch.no_grad():
            net(x)


# ---------------------------------------------------------------- mixed-precision TRAINING mode
GRAD_BOUND_MODEL = {"f16": 3e-2, "bf16": 4e-2}     # vs autograd of the quantised CPU model (sign flips)
GRAD_BOUND_STRICT = {"f16": 3e-3, "bf16": 2e-2}    # vs a manual backward on the kernel's OWN activations
DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def lp_inputs(P, seed=9):
    torch.manual_seed(seed)
    pts = torch.rand(P, 3) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)
    x = torch.cat([O.embed(pts, 9), vd], -1)
    G = torch.randn(P, 4) * torch.logspace(-5, 0, P)[:, None] * 1e-3       # mean-loss scale, 5 decades
    G[::7] = 0.0
    return x, G


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_training_gradients_vs_quantised_model(dev, prec):
    """train_precision = 'f16' / 'bf16': forward, dgrad and wgrad on 16-bit MFMAs.  Reference =
    autograd through tests/lp_reference.py (the oracle's NeRF.forward with the kernel's rounding
    points).  The residual is dominated by ReLU sign flips of units whose pre-activation is within
    the two implementations' 1e-4 forward difference of zero, hence the few-percent bound."""
    from lp_reference import nerf_forward_lp
    params = O.nerf_init(5)
    net = make_net(params, dev)
    net.train_precision = prec
    x, G = lp_inputs(3000)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    want = nerf_forward_lp(po, x, DT[prec])
    (want * G).sum().backward()
    out = net(x.to(dev))
    (out * G.to(dev)).sum().backward()
    assert rel_l2(out, want.detach()) < 1e-3
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert rel_l2(p.grad, po[k].grad) < GRAD_BOUND_MODEL[prec], (k, rel_l2(p.grad, po[k].grad))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_backward_on_the_kernels_own_activations(dev, prec):
    """Strict check of the dgrad chain and the weight-gradient kernel: the reference backward is
    evaluated in fp64 on the activations / sign pattern the 16-bit forward actually saved
    (workspace layout: include/scade_hip.h, scade_mlp_fwd_lp), so only the backward's own
    16-bit rounding of dZ remains."""
    from scade_amd import ops
    params = O.nerf_init(6)
    net = make_net(params, dev)
    bf16 = prec == "bf16"
    dt = DT[prec]
    P = 1500
    x, G = lp_inputs(P, seed=4)
    acts = ops.mlp_acts_lp_alloc(P, dev)
    ps = net.ordered_params()
    ops.mlp_fwd_lp(net.packed_lp(bf16), bf16, x.to(dev), None, None, acts)
    flat = ops.mlp_bwd_lp(None, net.packed_t_lp(bf16), bf16, acts, G.to(dev))
    torch.cuda.synchronize()
    raw = acts.cpu()
    slots = raw[:10 * P * 256 * 2].view(dt).view(10, P, 256).double()
    emb = raw[10 * P * 256 * 2:(10 * P * 256 + P * 64) * 2].view(dt).view(P, 64).double()
    a_off = ((10 * P * 256 + P * 64) * 2 + 255) // 256 * 256
    alpha_pre = raw[a_off:a_off + 4 * P].view(torch.float32).double()
    q = lambda k: params[k].to(dt).double()
    Gd = G.double()
    gam, view = emb[:, :57], emb[:, 60:63]
    want = {}
    hv = slots[8][:, :128]
    want["rgb_linear.weight"] = Gd[:, :3].T @ hv
    want["rgb_linear.bias"] = Gd[:, :3].sum(0)
    dzv = (Gd[:, :3] @ params["rgb_linear.weight"].double()) * (hv > 0)
    want["views_linears.0.weight"] = dzv.T @ torch.cat([slots[9], view], -1)
    want["views_linears.0.bias"] = dzv.sum(0)
    dfeat = dzv @ q("views_linears.0.weight")[:, :256]
    want["feature_linear.weight"] = dfeat.T @ slots[7]
    want["feature_linear.bias"] = dfeat.sum(0)
    dal = Gd[:, 3] * torch.sigmoid(10 * alpha_pre)
    want["alpha_linear.weight"] = (dal[:, None] * slots[7]).sum(0, keepdim=True)
    want["alpha_linear.bias"] = dal.sum().reshape(1)
    dh = dfeat @ q("feature_linear.weight") + dal[:, None] * params["alpha_linear.weight"].double()
    for l in range(7, -1, -1):
        dz = dh * (slots[l] > 0)
        inp = gam if l == 0 else (torch.cat([gam, slots[4]], -1) if l == 5 else slots[l - 1])
        want[f"pts_linears.{l}.weight"] = dz.T @ inp
        want[f"pts_linears.{l}.bias"] = dz.sum(0)
        if l > 0:
            Wq = q(f"pts_linears.{l}.weight")
            dh = dz @ (Wq[:, 57:] if l == 5 else Wq)
    o = 0
    for name in ops.PARAM_ORDER:
        n = want[name].numel()
        got = flat[o:o + n].cpu().double().view(want[name].shape)
        o += n
        assert rel_l2(got, want[name]) < GRAD_BOUND_STRICT[prec], (name, rel_l2(got, want[name]))
    assert o == flat.numel()


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_backward_power_of_two_scale_invariance_and_chunks(dev, prec):
    """Loss scaling is by exact powers of two (per point in the dgrad chain, per launch for the
    stored dZ): multiplying the upstream gradient by 2^-20 must scale every result by exactly 2^-20.
    Run at a size that spans several weight-gradient chunks and a ragged last tile."""
    params = O.nerf_init(7)
    net = make_net(params, dev)
    net.train_precision = prec
    x, G = lp_inputs(4001, seed=2)
    xd = x.to(dev)

    def grads(Gs):
        net.zero_grad(set_to_none=True)
        (net(xd) * Gs.to(dev)).sum().backward()
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()])

    g1, g2, g3 = grads(G), grads(G * 2.0 ** -20), grads(G)
    assert torch.equal(g1, g3)                                     # deterministic
    assert torch.equal(g1, g2 * 2.0 ** 20)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0


@pytest.mark.parametrize("prec", ["f16", "bf16", "bf16-s8"])
def test_lp_trainer_descends_like_the_exact_path(dev, prec):
    """Twelve optimiser steps of the full SCADE train step (render, 3-term loss with K hypotheses,
    backward, fused Adam, re-pack) in mixed precision track the exact fp32 run."""
    from scade_amd.train import Trainer, make_scade_nets
    N, K = 256, 20
    rays = O.synthetic_rays(N, seed=1).to(dev)
    torch.manual_seed(1)
    tgt = torch.rand(N, 3, device=dev) * 0.2 + 0.4
    hyp = torch.rand(K, N, 1, device=dev) * 4.9 + 0.1
    g = torch.Generator().manual_seed(3)
    draws = [(torch.rand(N, 64, generator=g).to(dev), torch.rand(N, 128, generator=g).to(dev),
              torch.rand(N, 128, generator=g).to(dev)) for _ in range(12)]
    runs = {}
    for p in ("f32", prec):
        coarse, fine = make_scade_nets(dev, seed=0)
        tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=p)
        runs[p] = [float(tr.step(rays, tgt, hyp, t_rand=a, u_coarse=b, cached_u=c)[0]) for a, b, c in draws]
    exact, lp = runs["f32"], runs[prec]
    assert all(v == v for v in lp)
    assert lp[-1] < 0.7 * lp[0], lp
    for a, b in zip(lp, exact):
        assert abs(a - b) <= 0.05 * abs(b), (lp, exact)


def test_lp_full_size_properties(dev):
    """BASELINE size (196,608 points = 1024 rays x 192, 85 weight-gradient chunks): the bf16 forward
    is blind to a permutation of the points, the backward is deterministic, exactly homogeneous under
    a power-of-two rescaling of the upstream gradient, and close to the exact fp32 backward."""
    params = O.nerf_init(2)
    net = make_net(params, dev)
    torch.manual_seed(0)
    P = 1024 * 192
    pts = torch.rand(P, 3) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1)
    x = torch.cat([O.embed(pts, 9), vd], -1).to(dev)
    G = torch.randn(P, 4, device=dev) * 1e-5
    perm = torch.randperm(P, device=dev)
    net.inference_precision = "bf16"
    with torch.no_grad():
        a = net(x)
        b = net(x[perm])
    assert torch.equal(a[perm], b)

    def grads(prec, Gs):
        net.train_precision = prec
        net.zero_grad(set_to_none=True)
        (net(x) * Gs).sum().backward()
        return torch.cat([p.grad.reshape(-1) for p in net.parameters()])

    g1, g2, g3 = grads("bf16", G), grads("bf16", G), grads("bf16", G * 2.0 ** -12)
    assert torch.equal(g1, g2)
    assert torch.equal(g1, g3 * 2.0 ** 12)
    exact = grads("f32", G)
    assert rel_l2(g1, exact) < 0.15, rel_l2(g1, exact)          # bf16 rounding + ReLU sign flips
    assert rel_l2(grads("f16", G), exact) < 0.06


@pytest.mark.parametrize("prec", ["bf16", "f16", "bf16-s8"])
def test_lp_training_curve_parity(dev, prec):
    """Loss-curve parity (the acceptance criterion for the config-5 path): a
    student pair of networks is fitted for 200 optimiser steps to a teacher's render (target colours
    and depth hypotheses around the teacher's depth) once with the exact fp32 kernels and once in
    mixed precision, identical draws: the curves stay together and the final PSNRs agree."""
    from scade_amd.train import Trainer, make_scade_nets
    N, K, steps = 512, 20, 200
    rays = O.synthetic_rays(N, seed=21).to(dev)
    tc, tf = make_scade_nets(dev, seed=100)                          # teacher
    e, _ = S.get_embedder(9, 0)
    ed, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(e, ed, torch.zeros(3, device=dev), torch.tensor(0.2, device=dev))
    with torch.no_grad():
        t = S.render_rays(rays, True, tc, query, 64, N_importance=128, network_fine=tf, perturb=0.)
    tgt = t["rgb_map"].clone()
    g = torch.Generator().manual_seed(22)
    hyp = (t["depth_map"][None, :, None] + 0.3 * torch.randn(K, N, 1, generator=g).to(dev)).clamp(0.1, 5.0)
    draws = [(torch.rand(N, 64, generator=g).to(dev), torch.rand(N, 128, generator=g).to(dev),
              torch.rand(N, 128, generator=g).to(dev)) for _ in range(steps)]
    curves = {}
    for p in ("f32", prec):
        coarse, fine = make_scade_nets(dev, seed=7)
        tr = Trainer(coarse, fine, torch.zeros(3), torch.tensor(0.2), n_images=1, precision=p)
        out = []
        for a, b, c in draws:
            loss, aux = tr.step(rays, tgt, hyp, t_rand=a, u_coarse=b, cached_u=c)
            out.append((float(loss), float(aux["img_loss"])))
        curves[p] = out
    exact, lp = curves["f32"], curves[prec]
    psnr_e = -10 * torch.log10(torch.tensor([x[1] for x in exact[-20:]]).mean())
    psnr_l = -10 * torch.log10(torch.tensor([x[1] for x in lp[-20:]]).mean())
    print(f"final PSNR (mean of last 20 steps): exact {float(psnr_e):.2f} dB, {prec} {float(psnr_l):.2f} dB; "
          f"loss {exact[0][0]:.4f} -> {exact[-1][0]:.5f} / {lp[-1][0]:.5f}")
    assert exact[-1][0] < 0.2 * exact[0][0] and lp[-1][0] < 0.2 * lp[0][0]      # both actually train
    assert abs(float(psnr_e) - float(psnr_l)) < 0.5
    mean_e = sum(x[0] for x in exact) / steps
    mean_l = sum(x[0] for x in lp) / steps
    assert abs(mean_e - mean_l) < 0.1 * mean_e


@pytest.mark.parametrize("prec", ["f16", "bf16", "f16x3"])
def test_reduced_precision_backward_survives_denormal_gradient_rows(dev, prec):
    """Regression (found by a 3000-step soak run): upstream gradient rows of denormal magnitude
    (samples whose compositing weight has decayed to ~1e-40) used to turn the per-point power-of-two
    scale into inf and poison the parameters with NaN.  The scales are clamped now; such rows are
    simply too small to contribute."""
    params = O.nerf_init(8)
    net = make_net(params, dev)
    x, G = lp_inputs(1500, seed=3)
    G = G.clone()
    G[5::11] = G[5::11].sign() * 1e-41           # denormal rows
    G[7::13] = 0.0
    G[3::17, :3] = 0.0
    G[3::17, 3] = 1e-44                          # denormal alpha gradient only
    res = {}
    for p in ("f32", prec):
        net.train_precision = p
        net.zero_grad(set_to_none=True)
        (net(x.to(dev)) * G.to(dev)).sum().backward()
        res[p] = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    assert torch.isfinite(res[prec]).all()
    bound = {"f16": 0.06, "bf16": 0.15, "f16x3": 1e-4}[prec]
    assert rel_l2(res[prec], res["f32"]) < bound, rel_l2(res[prec], res["f32"])
    # and an all-denormal batch (launch-wide maximum itself denormal)
    net.train_precision = prec
    net.zero_grad(set_to_none=True)
    (net(x.to(dev)) * torch.full_like(G, 1e-42).to(dev)).sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in net.parameters())


def test_bf16_backward_without_point_scale_flushes_only_negligible_gradients(dev):
    """The bf16 dgrad chain runs without the per-point power-of-two scale (mlp_bwd_lp.hip: S = s_p = 1 is
    bit-identical while every value stays in bf16's NORMAL range).  The caveat made explicit: rows whose
    output gradient is ~1e-36 lose their dZ values to the subnormal flush - the result stays finite, those
    rows contribute (at most) what the exact path gives them, and the rows of ordinary magnitude in the same
    batch are not disturbed."""
    params = O.nerf_init(8)
    net = make_net(params, dev)
    x, G = lp_inputs(1024, seed=5)
    tiny = torch.zeros(1024, dtype=torch.bool)
    tiny[::2] = True
    G_mixed = G.clone()
    G_mixed[tiny] = G_mixed[tiny].sign() * 1e-36
    G_big = G.clone()
    G_big[tiny] = 0.0
    out = {}
    for name, p, g_ in (("mixed", "bf16", G_mixed), ("big_only", "bf16", G_big), ("mixed_exact", "f32", G_mixed)):
        net.train_precision = p
        net.zero_grad(set_to_none=True)
        (net(x.to(dev)) * g_.to(dev)).sum().backward()
        out[name] = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    assert torch.isfinite(out["mixed"]).all()
    # the tiny rows change the bf16 weight gradients by no more than their (1e-36-sized) contribution allows:
    # compare the two bf16 runs - same rows of ordinary magnitude, same rounding
    # (to fp32 rounding of the sums: one ulp of a gradient entry at most, measured 4.5e-13)
    assert float((out["mixed"] - out["big_only"]).abs().max()) <= 1e-6 * float(out["big_only"].abs().max()), \
        float((out["mixed"] - out["big_only"]).abs().max())
    assert rel_l2(out["mixed"], out["mixed_exact"]) < 0.15


@pytest.mark.parametrize("P", [3000, 40000])
def test_bf16_with_8bit_saved_rows(dev, P):
    """train_precision = "bf16-s8" (format code 2 of the scade_mlp_*_lp entries): bf16 arithmetic with the rows
    saved for the weight gradient - activations written by the forward, dZ rows written by the dgrad chain - kept
    as 8-bit e5m2 in HBM (half the bytes of the HBM-bound 16-bit training step).  Checked here:
      * the forward's OUTPUT is the bf16 path's, bit for bit (only the tile copies are rounded);
      * every saved activation row IS the bf16 path's row rounded to e5m2 (RNE), bit for bit;
      * the weight gradients equal dZ8^T In8 / S evaluated in fp64 on the kernel's OWN 8-bit rows (S = the
        launch-wide power-of-two loss scale) to fp32-accumulation accuracy - the kernels compute exactly the
        contraction they are given;
      * against the plain bf16 gradient the difference is the e5m2 rounding of the operands: a few per cent
        norm-wise, zero-mean.
    3000 points run 64-point workgroups, 40,000 points 128-point ones."""
    import math
    from scade_amd import ops, _lib
    from scade_amd._lib import call, ptr, stream
    params = O.nerf_init(5)
    net = make_net(params, dev)
    x, G = lp_inputs(P, seed=11)
    x, G = x.to(dev), G.to(dev)
    lib = _lib.load()
    res = {}
    for code in (1, 2):
        acts = ops.mlp_acts_lp_alloc(P, dev)
        acts.zero_()
        out = ops.mlp_fwd_lp(net.packed_lp(True), code, x, None, None, acts)
        ws = torch.zeros(int(lib.scade_mlp_bwd_lp_workspace_bytes(P)), device=dev, dtype=torch.uint8)
        grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
        call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(True)), code, ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
        torch.cuda.synchronize()
        res[code] = (out, acts, ws, grad)
    (o16, a16, w16, g16), (o8, a8, w8, g8) = res[1], res[2]
    assert torch.equal(o16, o8), "the forward arithmetic must not change"
    rows16 = a16[:10 * P * 512].view(torch.bfloat16).view(10, P, 256)
    slot8 = lambda t, s: t[s * P * 512:s * P * 512 + P * 256].view(torch.float8_e5m2).view(P, 256)
    for s_ in range(10):
        if s_ == 8:                                                 # the 128-wide views hidden layer stays 16-bit
            rows8 = a8[:10 * P * 512].view(torch.bfloat16).view(10, P, 256)[8][:, :128]
            assert torch.equal(rows8, rows16[8][:, :128]), "views hidden slot (16-bit in both formats)"
            continue
        want = rows16[s_].to(torch.float8_e5m2)
        assert torch.equal(slot8(a8, s_).view(torch.uint8), want.view(torch.uint8)), f"activation slot {s_}"
    # the embedding rows (read by the weight gradient only): fp8 e4m3 in format code 2, 64 bytes per point
    e0 = 10 * P * 512
    emb16 = a16[e0:e0 + P * 128].view(torch.bfloat16).view(P, 64)
    emb8 = a8[e0:e0 + P * 64].view(torch.float8_e4m3fn).view(P, 64)
    assert torch.equal(emb8.view(torch.uint8), emb16.to(torch.float8_e4m3fn).view(torch.uint8)), "embedding rows"
    assert rel_l2(emb8.float(), emb16.float()) < 0.04 and float(emb16.float().abs().max()) <= 1.0
    # the kernel's own rows -> the contraction in fp64
    # (the launch-wide scale is taken from the EFFECTIVE output gradient: the density channel behind the softplus'
    # derivative sigmoid(10 alpha_pre) - mlp_bwd_lp.hip lp_effective_g3)
    a_off = ((10 * P * 256 + P * 64) * 2 + 255) // 256 * 256
    alpha_pre = a8[a_off:a_off + 4 * P].view(torch.float32)
    eff = G.clone()
    eff[:, 3] = torch.where(10 * alpha_pre > 20, G[:, 3], G[:, 3] / (1 + torch.exp(-10 * alpha_pre)))
    m = float(eff.abs().max())
    S = 2.0 ** min(6 - math.frexp(m)[1], 96)
    off, flat = 0, {}
    for name in ops.PARAM_ORDER:
        n = math.prod(ops.PARAM_SHAPES[name])
        flat[name] = (g8[off:off + n].view(ops.PARAM_SHAPES[name]), g16[off:off + n].view(ops.PARAM_SHAPES[name]))
        off += n
    for l in range(1, 8):
        dz = slot8(w8, l).double()
        inp = slot8(a8, l - 1).double()
        ref = dz.t() @ inp / S
        got, plain = flat[f"pts_linears.{l}.weight"]
        got = got[:, 57:] if l == 5 else got
        plain = plain[:, 57:] if l == 5 else plain
        assert rel_l2(got, ref) < 2e-3, (l, rel_l2(got, ref))
        assert rel_l2(flat[f"pts_linears.{l}.bias"][0], dz.sum(0) / S) < 2e-3
        assert rel_l2(got, plain) < 0.08, (l, rel_l2(got, plain))       # e5m2 rounding of both operands
    # the embedding-input jobs (bf8 x fp8 operands): layer 0, the skip block of layer 5, the view columns of the views layer
    e8 = emb8.double()
    for name, dslot, cols, kcols in (("pts_linears.0.weight", 0, slice(0, 57), slice(0, 57)),
                                     ("pts_linears.5.weight", 5, slice(0, 57), slice(0, 57)),
                                     ("views_linears.0.weight", 8, slice(256, 259), slice(60, 63))):
        dz = slot8(w8, dslot).double()
        dz = dz[:, :128] if dslot == 8 else dz
        ref = dz.t() @ e8[:, kcols] / S
        got, plain = flat[name]
        assert rel_l2(got[:, cols], ref) < 2e-3, (name, rel_l2(got[:, cols], ref))
        assert rel_l2(got[:, cols], plain[:, cols]) < 0.08, (name, rel_l2(got[:, cols], plain[:, cols]))
    assert rel_l2(flat["pts_linears.0.bias"][0], slot8(w8, 0).double().sum(0) / S) < 2e-3
    # the feature layer's job with the alpha-head rider (d alpha_pre fp32 x the 8-bit rows of slot 7), and the views layer
    dzf, h7 = slot8(w8, 9).double(), slot8(a8, 7).double()
    assert rel_l2(flat["feature_linear.weight"][0], dzf.t() @ h7 / S) < 2e-3
    assert rel_l2(flat["feature_linear.bias"][0], dzf.sum(0) / S) < 2e-3
    dal = w8[10 * P * 512:].view(torch.float32)[:P].double()          # lp_dz_dalpha_byte(P): 256-byte aligned already
    assert rel_l2(flat["alpha_linear.weight"][0].reshape(-1), dal @ h7) < 2e-3
    assert rel_l2(flat["alpha_linear.bias"][0].reshape(-1), dal.sum().reshape(1)) < 2e-3
    dzv = slot8(w8, 8).double()[:, :128]
    assert rel_l2(flat["views_linears.0.weight"][0][:, :256], dzv.t() @ slot8(a8, 9).double() / S) < 2e-3
    assert rel_l2(flat["views_linears.0.bias"][0], dzv.sum(0) / S) < 2e-3
    # the dgrad chain itself is the bf16 path's: its dZ rows are the bf16 rows (x S) rounded to e5m2
    dz16 = w16[:10 * P * 512].view(torch.bfloat16).view(10, P, 256)
    for s_ in (1, 4, 7, 9):
        want = (dz16[s_].float() * S).to(torch.float8_e5m2).float()
        got8 = slot8(w8, s_).float()
        normal = want.abs() >= 2.0 ** -14
        assert torch.equal(got8[normal], want[normal]), f"dZ slot {s_}: normal range"
        # (e5m2's subnormal range, 2^-16 steps: the hardware conversion and torch's may treat it differently)
        assert float((got8 - want).abs().max()) <= 2.0 ** -16, f"dZ slot {s_}: subnormal range"
        assert float(normal.float().mean()) > 0.3
    # all 24 tensors: finite, close to the plain bf16 gradient norm-wise
    # (measured 6.5 - 7 % on this input, whose upstream gradients span five decades: a few points dominate every
    # sum, so the zero-mean e5m2 rounding of the operands hardly averages out; uniform batches average far more)
    assert torch.isfinite(g8).all()
    assert rel_l2(g8, g16) < 0.09, rel_l2(g8, g16)


def test_bf16_s8_weight_gradient_at_config5_size(dev):
    """The 8-bit weight gradient at the FULL size of BASELINE config 5's fine launch (4096 rays x 192 samples =
    786,432 points): the balanced persistent plan cuts every job into segments per workgroup there, and a segment
    boundary that dropped or doubled a stage of 64 points would pass every small-P test.  One ring job (a hidden
    layer: pts_linears.3) and the embedding job (pts_linears.0, bf8 x fp8 operands) against the fp64 contraction of
    the kernel's OWN 8-bit rows, as test_bf16_with_8bit_saved_rows does at small P; every tensor finite."""
    import math
    from scade_amd import ops, _lib
    from scade_amd._lib import call, ptr, stream
    P = 4096 * 192
    params = O.nerf_init(5)
    net = make_net(params, dev)
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.cat([torch.rand(P, 3, device=dev, generator=g) * 2 - 1, torch.randn(P, 3, device=dev, generator=g)], -1)
    x = torch.cat([S.get_embedder(9, 0)[0](x[:, :3]), torch.nn.functional.normalize(x[:, 3:], dim=-1)], -1).contiguous()
    G = torch.randn(P, 4, device=dev, generator=g) * 1e-4
    lib = _lib.load()
    acts = ops.mlp_acts_lp_alloc(P, dev)
    ops.mlp_fwd_lp(net.packed_lp(True), 2, x, None, None, acts)
    ws = torch.zeros(int(lib.scade_mlp_bwd_lp_workspace_bytes(P)), device=dev, dtype=torch.uint8)
    grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
    call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(True)), 2, ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
    torch.cuda.synchronize()
    assert torch.isfinite(grad).all()
    slot8 = lambda t, s: t[s * P * 512:s * P * 512 + P * 256].view(torch.float8_e5m2).view(P, 256)
    S_ = 2.0 ** min(6 - math.frexp(float(G.abs().max()))[1], 96)
    off, flat = 0, {}
    for name in ops.PARAM_ORDER:
        n = math.prod(ops.PARAM_SHAPES[name])
        flat[name] = grad[off:off + n].view(ops.PARAM_SHAPES[name])
        off += n

    def contract(dz8, in8):
        """dz^T in in fp64, in slices of 65,536 points (a [P, 256] fp64 copy of both operands is 3 GB)"""
        acc = torch.zeros(dz8.shape[1], in8.shape[1], device=dev, dtype=torch.float64)
        col = torch.zeros(dz8.shape[1], device=dev, dtype=torch.float64)
        for a in range(0, P, 65536):
            d = dz8[a:a + 65536].double()
            acc += d.t() @ in8[a:a + 65536].double()
            col += d.sum(0)
        return acc / S_, col / S_
    ref_w, ref_b = contract(slot8(ws, 3), slot8(acts, 2))
    assert rel_l2(flat["pts_linears.3.weight"], ref_w) < 2e-3, rel_l2(flat["pts_linears.3.weight"], ref_w)
    assert rel_l2(flat["pts_linears.3.bias"], ref_b) < 2e-3
    e0 = 10 * P * 512
    emb8 = acts[e0:e0 + P * 64].view(torch.float8_e4m3fn).view(P, 64)
    ref_e, ref_b0 = contract(slot8(ws, 0), emb8)
    assert rel_l2(flat["pts_linears.0.weight"], ref_e[:, :57]) < 2e-3, rel_l2(flat["pts_linears.0.weight"], ref_e[:, :57])
    assert rel_l2(flat["pts_linears.0.bias"], ref_b0) < 2e-3


@pytest.mark.parametrize("P", [1, 65, 200, 3001])
@pytest.mark.parametrize("fmt", ["f32", "bf16", "bf16-s8"])
def test_ragged_tiles_stay_inside_their_slots(dev, fmt, P):
    """Canary bytes around everything the training kernels write: the tile copies of the forward and the
    dgrad chain are buffer stores whose ragged last tile is cut by the descriptor's range check (rows >= P are dropped,
    no branch), and an 8-bit row uses the first half of its 16-bit slot - an overrun into the unused half, or past the
    end of the buffer, would go unnoticed by every parity test.  Saved-row buffer and backward workspace are allocated
    with a 64 KiB tail and pre-filled with 0xA5; after forward + backward the tail is intact, and in format code 2 so
    is the unused half of every slot (activations, dZ) and of the embedding rows."""
    from scade_amd import ops, _lib
    from scade_amd._lib import call, ptr, stream
    lib = _lib.load()
    params = O.nerf_init(5)
    net = make_net(params, dev)
    x, G = lp_inputs(P, seed=31)
    x, G = x.to(dev).contiguous(), G.to(dev).contiguous()
    PAD = 65536
    if fmt == "f32":
        na, nw = int(lib.scade_mlp_acts_floats(P)) * 4, int(lib.scade_mlp_bwd_workspace_floats(P)) * 4
    else:
        na, nw = int(lib.scade_mlp_acts_lp_bytes(P)), int(lib.scade_mlp_bwd_lp_workspace_bytes(P))
    acts = torch.full((na + PAD,), 0xA5, device=dev, dtype=torch.uint8)
    ws = torch.full((nw + PAD,), 0xA5, device=dev, dtype=torch.uint8)
    grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
    out = torch.empty(P, 4, device=dev)
    if fmt == "f32":
        call("scade_mlp_fwd", ptr(net.packed()), 0, ptr(x), None, 0, None, P, 1, ptr(out), ptr(acts), stream())
        call("scade_mlp_bwd", ptr(net.packed()), ptr(net.packed_t()), ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
    else:
        code = 2 if fmt == "bf16-s8" else 1
        out = ops.mlp_fwd_lp(net.packed_lp(True), code, x, None, None, acts[:na])
        call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(True)), code, ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(grad).all()
    assert bool((acts[na:] == 0xA5).all()), "saved rows: bytes past the end of the buffer were written"
    assert bool((ws[nw:] == 0xA5).all()), "backward workspace: bytes past the end of the buffer were written"
    assert not bool((acts[:min(na, P * 256)] == 0xA5).all()), "(the forward did write its rows)"
    if fmt == "bf16-s8":
        for s_ in range(10):
            for name, buf in (("activation", acts), ("dZ", ws)):
                if s_ == 8 and name == "activation":
                    # the 128-wide views hidden layer stays 16-bit (512-byte row pitch, 128 columns = 256 bytes used)
                    unused = buf[s_ * P * 512:(s_ + 1) * P * 512].view(P, 512)[:, 256:]
                else:
                    unused = buf[s_ * P * 512 + P * 256:(s_ + 1) * P * 512]
                assert bool((unused == 0xA5).all()), f"{name} slot {s_}: the unused half of an 8-bit slot was written"
        e0 = 10 * P * 512
        assert bool((acts[e0 + P * 64:e0 + P * 128] == 0xA5).all()), "embedding rows: 64 fp8 bytes per point, no more"


def test_bf16_s8_points_outside_the_scene_box_keep_finite_fp8_rows(dev):
    """Format code 2 saves the embedding rows as fp8 e4m3 (|sin|, |cos|, |viewdir| <= 1) - but columns 0..2 are the raw
    normalised coordinate, which is unbounded for a point outside the bounding box.  e4m3 ends at 448: the
    saved coordinate is clamped there, stays finite, and so does every weight gradient; in-box points are untouched."""
    from scade_amd import ops, _lib
    from scade_amd._lib import call, ptr, stream
    P = 300
    net = make_net(O.nerf_init(5), dev)
    x, G = lp_inputs(P, seed=41)
    x[5, 0], x[9, 1], x[200, 2] = 1.0e4, -2.0e3, 449.0
    x, G = x.to(dev).contiguous(), G.to(dev).contiguous()
    acts = ops.mlp_acts_lp_alloc(P, dev)
    out = ops.mlp_fwd_lp(net.packed_lp(True), 2, x, None, None, acts)
    ws = torch.zeros(int(_lib.load().scade_mlp_bwd_lp_workspace_bytes(P)), device=dev, dtype=torch.uint8)
    grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
    call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(True)), 2, ptr(acts), ptr(G), P, ptr(ws), ptr(grad), stream())
    torch.cuda.synchronize()
    e0 = 10 * P * 512
    emb8 = acts[e0:e0 + P * 64].view(torch.float8_e4m3fn).view(P, 64).float()
    assert torch.isfinite(emb8).all()
    assert float(emb8[5, 0]) == 448.0 and float(emb8[9, 1]) == -448.0 and float(emb8[200, 2]) == 448.0
    inbox = torch.ones(P, dtype=torch.bool, device=dev)
    inbox[[5, 9, 200]] = False
    want = x[:, :3].to(torch.bfloat16).to(torch.float8_e4m3fn).float()
    assert torch.equal(emb8[inbox][:, :3], want[inbox])
    assert torch.isfinite(grad).all() and torch.isfinite(out[inbox]).all()


def test_loss_scale_of_the_8bit_rows_ignores_a_density_outlier(dev):
    """Round 6 (found by the full-size bucket comparison, tests/test_gpu_config5.py): the 8-bit dZ rows carry ONE
    power-of-two scale per launch, taken from max |d loss / d raw|.  The reference's arithmetic produces single entries
    of d loss / d sigma that are ten decades above the rest - the last sample's delta = 1e10 (run_scade_scannet.py:515),
    an importance sampler's empty bin (helpers:371) - at samples whose softplus derivative sigmoid(10 alpha_pre) is
    as small, so they carry nothing into the backward; taken raw, one of them set the scale, every e5m2 row of the fine
    network underflowed and its weight gradient came out as exact zeros.  The maximum is now taken over the effective
    gradient (the density channel behind its sigmoid): an outlier of 1e3 at a point with alpha_pre ~ -3 must leave the
    gradient where it was."""
    from scade_amd import ops, _lib
    from scade_amd._lib import call, ptr, stream
    params = O.nerf_init(5)
    params["alpha_linear.bias"] = torch.tensor([-3.0])             # alpha_pre ~ -3: sigmoid(10 alpha_pre) ~ 1e-13
    net = make_net(params, dev)
    P = 3000
    x, G = lp_inputs(P, seed=12)
    x, G = x.to(dev), G.to(dev)
    lib = _lib.load()
    grads = []
    for outlier in (False, True):
        Gx = G.clone()
        if outlier:
            Gx[7, 3] = 1.0e3
        acts = ops.mlp_acts_lp_alloc(P, dev)
        ops.mlp_fwd_lp(net.packed_lp(True), 2, x, None, None, acts)
        ws = torch.zeros(int(lib.scade_mlp_bwd_lp_workspace_bytes(P)), device=dev, dtype=torch.uint8)
        grad = torch.empty(ops.N_PARAM_FLOATS, device=dev)
        call("scade_mlp_bwd_lp", None, ptr(net.packed_t_lp(True)), 2, ptr(acts), ptr(Gx), P, ptr(ws), ptr(grad), stream())
        torch.cuda.synchronize()
        grads.append(grad)
    a_off = ((10 * P * 256 + P * 64) * 2 + 255) // 256 * 256
    assert float(acts[a_off:a_off + 4 * P].view(torch.float32)[7]) < -2.0, "the outlier sits at a sample with a dead softplus"
    g0, g1 = grads
    n_w = 256 * 57 + 256                                           # pts_linears.0: an MFMA-contracted tensor
    assert float(g0[:n_w].abs().max()) > 0 and float(g1[:n_w].abs().max()) > 0
    assert rel_l2(g1, g0) < 1e-3, rel_l2(g1, g0)
