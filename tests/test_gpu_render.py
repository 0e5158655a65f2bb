"""GPU parity of render_rays end-to-end: golden fixture from the real reference
(N=32), oracle on seeded inputs, and size-independent properties at the bench size."""
import pytest
import torch

import scade_amd as S
from conftest import assert_close, load_golden, rel_l2
from oracle import scade_oracle as O
from test_oracle_golden import f6_params
from test_gpu_ops import make_net

pytestmark = pytest.mark.gpu


def build(dev, pc, pf, bbc, bbs):
    coarse, fine = make_net(pc, dev), make_net(pf, dev)
    embed_fn, _ = S.get_embedder(9, 0)
    embeddirs_fn, _ = S.get_embedder(0, 0)
    query = S.make_network_query_fn(embed_fn, embeddirs_fn, bbc.to(dev), bbs.to(dev))
    return coarse, fine, query


KEYS_TIGHT = ["z_vals0", "z_vals", "u"]


def check_ret(ret, want, tag, rtol=1e-4):
    for k, v in want.items():
        assert k in ret, f"missing key {k}"
        atol = 1e-5
        if k in ("disp_map", "disp0"):
            atol = 1e-5 * float(v.abs().max())
        assert_close(ret[k], v, rtol=rtol, atol=atol, what=f"{tag} {k}")


def test_render_rays_det_golden(dev):
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    with torch.no_grad():
        ret = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, embedded_cam=torch.empty(0, device=dev),
                            N_importance=128, network_fine=fine, perturb=0., retraw=True)
    want = {k[4:]: v for k, v in g.items() if k.startswith("det/")}
    assert set(want) == set(ret)
    check_ret(ret, want, "det")
    assert torch.equal(ret["z_vals0"].cpu(), want["z_vals0"])


def test_render_rays_train_forward_golden(dev):
    g = load_golden("f6_render")
    pc, pf = f6_params(g)
    coarse, fine, query = build(dev, pc, pf, g["bb_center"], g["bb_scale"])
    with torch.no_grad():
        ret = S.render_rays(g["rays"].to(dev), True, coarse, query, 64, embedded_cam=torch.empty(0, device=dev),
                            N_importance=128, network_fine=fine, perturb=1., retraw=True, pytest=True)
        want = {k[6:]: v for k, v in g.items() if k.startswith("train/") and k[6:] in ret}
        check_ret(ret, want, "train-fwd")
        il = S.img2mse(ret["rgb_map"], g["target_s"].to(dev))
        cv = S.compute_space_carving_loss(ret["pred_hyp"], g["hyp"].to(dev))
        il0 = S.img2mse(ret["rgb0"], g["target_s"].to(dev))
    assert_close(il, g["train/img_loss"], rtol=1e-4, atol=1e-7, what="img_loss")
    assert_close(cv, g["train/carve"], rtol=1e-4, atol=1e-7, what="carve")
    assert_close(il0, g["train/img_loss0"], rtol=1e-4, atol=1e-7, what="img_loss0")


def test_render_rays_vs_oracle_seeded(dev):
    N = 96
    rays = O.synthetic_rays(N, seed=21)
    pc, pf = O.nerf_init(3), O.nerf_init(4)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    with torch.no_grad():
        want = O.render_rays(rays, pc, pf, bbc, bbs, retraw=True)
    coarse, fine, query = build(dev, pc, pf, bbc, bbs)
    with torch.no_grad():
        ret = S.render_rays(rays.to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                            perturb=0., retraw=True)
    check_ret(ret, want, "seeded")


def test_render_rays_full_size_properties(dev):
    """1024 rays x (64+128): structural invariants that do not need the oracle."""
    N = 1024
    rays = O.synthetic_rays(N, seed=0).to(dev)
    pc, pf = O.nerf_init(0), O.nerf_init(1)
    coarse, fine, query = build(dev, pc, pf, torch.zeros(3), torch.tensor(0.2))
    with torch.no_grad():
        ret = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                            perturb=1., retraw=True)
        ret2 = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                             perturb=0., retraw=True)
    for r in (ret, ret2):
        z = r["z_vals"]
        assert z.shape == (N, 192) and bool((z[:, 1:] >= z[:, :-1]).all()), "z_vals sorted"
        assert bool((r["weights"] >= 0).all()) and bool((r["acc_map"] <= 1 + 1e-5).all())
        assert_close(r["weights"].sum(-1), r["acc_map"], rtol=1e-5, atol=1e-6, what="acc == sum w")
        assert bool((r["pred_hyp"] >= z[:, :1]).all()) and bool((r["pred_hyp"] <= z[:, -1:]).all())
        assert not any(torch.isnan(v).any() for k, v in r.items() if k not in ("disp_map", "disp0"))
    # determinism of the det path
    with torch.no_grad():
        ret3 = S.render_rays(rays, True, coarse, query, 64, N_importance=128, network_fine=fine,
                             perturb=0., retraw=True)
    for k in ret2:
        assert torch.equal(torch.nan_to_num(ret2[k]), torch.nan_to_num(ret3[k])), k
    # chunking (batchify_rays) does not change results
    with torch.no_grad():
        all_ret = S.batchify_rays(rays, 300, True, network_fn=coarse, network_query_fn=query, N_samples=64,
                                  N_importance=128, network_fine=fine, perturb=0., retraw=True)
    for k in ret2:
        assert torch.equal(torch.nan_to_num(all_ret[k]), torch.nan_to_num(ret2[k])), k


def test_render_api_unsupported_paths_fail_loudly(dev):
    rays = O.synthetic_rays(8).to(dev)
    with pytest.raises(NotImplementedError):
        S.render_rays(rays, True, None, None, 64, N_importance=0)
