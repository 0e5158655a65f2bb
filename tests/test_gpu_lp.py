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
    assert rel_l2(fused, emb) < BOUND[prec] / 4      # only the sin/cos rounding differs


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lp_render_psnr_within_0p05_db(dev, prec):
    """North-star acceptance: PSNR of the render against a ground-truth image within 0.05 dB of
    the reference render's.  Ground truth = exact fp32 render (itself pinned to the oracle by
    tests/test_gpu_render.py) + noise at ~25 dB, a typical ScanNet test PSNR; 2048 rays so that
    the estimate is not dominated by the error-noise cross term."""
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    rays = torch.cat([g["rays"], O.synthetic_rays(2016, seed=11)], 0).to(dev)
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
