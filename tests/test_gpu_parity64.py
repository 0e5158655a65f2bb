"""Closes the end-to-end parity argument with a MEASUREMENT (run_scade_scannet.py:702-730).

After the coarse->fine resampling two correct fp32 implementations differ element-wise (sample
positions are an ill-conditioned function of the coarse weights; the 2^8 pi encoding amplifies a
1e-7 point shift ~800x), so the end-to-end tests compare norm-wise.  Here that is quantified: the
oracle is evaluated in fp64 (every operation in double on the same fp32 inputs, parameters and
draws = the exact-arithmetic answer to ~1e-15), and for every render_rays output and every
gradient tensor

    err(HIP fp32, fp64)  <=  C * err(oracle fp32 = the reference's own arithmetic, fp64)

i.e. the HIP path is as close to the true answer as the reference is; the remaining HIP-vs-oracle
difference is the fp32 noise floor of the problem, not an implementation error.  With
SCADE_PARITY_JSON=<path> the per-key errors are written out (committed as profiles/r02_parity.json).
"""
import json
import os

import pytest
import torch

import scade_amd as S
from conftest import rel_l2
from oracle import scade_oracle as O
from test_gpu_render import build

pytestmark = pytest.mark.gpu

C_BOUND = 2.0          # HIP may be at most this much farther from the fp64 answer than the reference's fp32
KEYS = ["rgb0", "disp0", "acc0", "depth0", "weights0", "z_vals0", "rgb_map", "disp_map", "acc_map", "depth_map",
        "z_vals", "weights", "pred_hyp", "z_std", "raw"]


class _fp64:
    def __enter__(self):
        self.prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *a):
        torch.set_default_dtype(self.prev)


def _errs(x, ref):
    """(rel-L2, 99.9th-percentile |d| / max|ref|, max |d| / max|ref|) against the fp64 answer."""
    x = torch.nan_to_num(x.detach().double().cpu()).flatten()
    r = torch.nan_to_num(ref.detach().double().cpu()).flatten()
    d = (x - r).abs()
    scale = float(r.abs().max()) + 1e-300
    q = float(torch.quantile(d[:: max(1, d.numel() // 2_000_000)], 0.999)) if d.numel() > 1 else float(d.max())
    return float(d.norm() / (r.norm() + 1e-300)), q / scale, float(d.max()) / scale


def _record(table, key, hip, orc, ref):
    eh, eo = _errs(hip, ref), _errs(orc, ref)
    table[key] = {"hip_rel_l2": eh[0], "oracle32_rel_l2": eo[0], "hip_p999": eh[1], "oracle32_p999": eo[1],
                  "hip_max": eh[2], "oracle32_max": eo[2], "hip_vs_oracle32_rel_l2": rel_l2(hip, orc)}
    return eh, eo


def _dump(section, table):
    path = os.environ.get("SCADE_PARITY_JSON")
    if not path:
        return
    data = {}
    if os.path.exists(path):
        data = json.load(open(path))
    data[section] = table
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def _check(table, floor):
    bad = []
    for k, r in table.items():
        # rel-L2 and the 99.9th percentile are statistics; the max over a few 1e5 elements is one
        # discrete event (an index flip next to a cdf knot) and is reported, not bounded
        for m in ("rel_l2", "p999"):
            if r["hip_" + m] > C_BOUND * r["oracle32_" + m] + floor:
                bad.append((k, m, r["hip_" + m], r["oracle32_" + m]))
    assert not bad, "HIP farther from the fp64 answer than C x the reference's own fp32: %s" % bad


def test_render_rays_error_against_fp64_is_the_references_own(dev):
    N = 512
    rays = O.synthetic_rays(N, seed=91)
    pc, pf = O.nerf_init(92), O.nerf_init(93)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    g = torch.Generator().manual_seed(94)
    t_rand, uc, uf = torch.rand(N, 64, generator=g), torch.rand(N, 128, generator=g), torch.rand(N, 128, generator=g)
    table = {}
    for tag, kw_o, kw_h in (("det", {}, dict(perturb=0.)),
                            ("train", dict(t_rand=t_rand, u_coarse=uc, u_fine=uf),
                             dict(perturb=1., t_rand=t_rand.to(dev), u_coarse=uc.to(dev), cached_u=uf.to(dev)))):
        with torch.no_grad():
            o32 = O.render_rays(rays, pc, pf, bbc, bbs, retraw=True, **kw_o)
            with _fp64():
                d = lambda t: t.double()
                o64 = O.render_rays(d(rays), {k: d(v) for k, v in pc.items()}, {k: d(v) for k, v in pf.items()},
                                    d(bbc), d(bbs), retraw=True, **{k: d(v) for k, v in kw_o.items()})
            coarse, fine, query = build(dev, pc, pf, bbc, bbs)
            hip = S.render_rays(rays.to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine,
                                retraw=True, **kw_h)
        sub = {}
        for k in KEYS:
            _record(sub, k, hip[k], o32[k], o64[k])
        psnr = lambda a, b: float(-10 * torch.log10(torch.mean((a.double().cpu() - b.double().cpu()) ** 2) + 1e-300))
        sub["_psnr_rgb_map_dB"] = {"hip_vs_fp64": psnr(hip["rgb_map"], o64["rgb_map"]),
                                   "oracle32_vs_fp64": psnr(o32["rgb_map"], o64["rgb_map"]),
                                   "hip_vs_oracle32": psnr(hip["rgb_map"], o32["rgb_map"])}
        table[tag] = sub
        _check({k: v for k, v in sub.items() if not k.startswith("_")}, floor=2e-7)
    _dump("render_rays_512rays", table)


def _train_step_grads(dev, N, K, seed, precision="f32"):
    """All 48 parameter gradients + scale/shift + loss of the 3-term loss (:968-985) for one seeded problem:
    (HIP, oracle fp32 = the reference's arithmetic, oracle fp64)."""
    rays = O.synthetic_rays(N, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    tgt = torch.rand(N, 3, generator=g)
    hyp = torch.rand(K, N, 1, generator=g) * 4.9 + 0.1
    t_rand, uc, uf = torch.rand(N, 64, generator=g), torch.rand(N, 128, generator=g), torch.rand(N, 128, generator=g)
    pc0, pf0 = O.nerf_init(seed + 2), O.nerf_init(seed + 3)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)

    def oracle_grads(cast):
        pc = {k: cast(v).clone().requires_grad_(True) for k, v in pc0.items()}
        pf = {k: cast(v).clone().requires_grad_(True) for k, v in pf0.items()}
        sc, sh = cast(torch.ones(1)).requires_grad_(True), cast(torch.zeros(1)).requires_grad_(True)
        ret = O.render_rays(cast(rays), pc, pf, cast(bbc), cast(bbs), t_rand=cast(t_rand), u_coarse=cast(uc),
                            u_fine=cast(uf))
        loss = O.train_loss(ret, cast(tgt), cast(hyp) * sc + sh)[0]
        loss.backward()
        zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
        out = {"coarse." + k: zero(v) for k, v in pc.items()}
        out.update({"fine." + k: zero(v) for k, v in pf.items()})
        out["depth_scale"], out["depth_shift"], out["loss"] = sc.grad, sh.grad, loss.detach()
        return out

    g32 = oracle_grads(lambda t: t.float())
    with _fp64():
        g64 = oracle_grads(lambda t: t.double())
    coarse, fine, query = build(dev, pc0, pf0, bbc, bbs)
    coarse.train_precision = fine.train_precision = precision
    sc = torch.ones(1, device=dev, requires_grad=True)
    sh = torch.zeros(1, device=dev, requires_grad=True)
    ret = S.render_rays(rays.to(dev), True, coarse, query, 64, N_importance=128, network_fine=fine, perturb=1.,
                        t_rand=t_rand.to(dev), u_coarse=uc.to(dev), cached_u=uf.to(dev))
    loss = S.img2mse(ret["rgb_map"], tgt.to(dev)) + 0.007 * S.compute_space_carving_loss(
        ret["pred_hyp"], hyp.to(dev) * sc + sh) + S.img2mse(ret["rgb0"], tgt.to(dev))
    loss.backward()
    hipg = {"coarse." + k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in coarse.named_parameters()}
    hipg.update({"fine." + k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in fine.named_parameters()})
    hipg["depth_scale"], hipg["depth_shift"], hipg["loss"] = sc.grad, sh.grad, loss.detach()
    return hipg, g32, g64


# gradient tensors with fewer than 16 elements (depth scale / shift, the alpha head's bias, the rgb head's
# bias): ONE seeded problem gives one error draw per element, not a statistic - they are bounded over many
# seeds by test_small_gradient_tensors_error_is_the_references_own_over_16_seeds below
def _is_small(t):
    return t.numel() < 16


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_train_step_gradient_error_against_fp64_is_the_references_own(dev, precision):
    """All 48 parameter gradients + scale/shift of the 3-term loss (:968-985).  "f16x3": the split-precision training
    kernels, whose saved rows are 3 bytes per value since round 5 (fp16 h + e5m2 l: the weight gradient's operands to
    14 significant bits) - held to the SAME criterion as the exact path."""
    hipg, g32, g64 = _train_step_grads(dev, 256, 20, seed=95, precision=precision)
    table = {}
    for k in g64:
        if float(g64[k].abs().max()) == 0.0:
            assert float(hipg[k].abs().max()) == 0.0, f"{k}: must be exactly zero"
            continue
        _record(table, k, hipg[k], g32[k], g64[k])
    cat = lambda d, pre: torch.cat([d[k].detach().double().cpu().flatten() for k in g64 if k.startswith(pre)])
    for pre in ("coarse.", "fine."):
        _record(table, "_all_" + pre[:-1], cat(hipg, pre), cat(g32, pre), cat(g64, pre))
    _dump("train_step_grads_256rays_K20" + ("" if precision == "f32" else "_" + precision), table)
    # a single gradient tensor's max/percentile is dominated by a handful of near-knot samples;
    # bound the norm-wise error of every tensor and both statistics of the whole networks.  The few-element
    # tensors are recorded here and BOUNDED by the multi-seed test (no absolute floor any more).
    small = {k for k in table if not k.startswith("_") and _is_small(g64[k])}
    bad = [(k, r["hip_rel_l2"], r["oracle32_rel_l2"]) for k, r in table.items()
           if k not in small and r["hip_rel_l2"] > C_BOUND * r["oracle32_rel_l2"] + 2e-7]
    assert not bad, "HIP gradients farther from fp64 than C x the reference's fp32: %s" % bad
    for k in ("_all_coarse", "_all_fine"):
        assert table[k]["hip_p999"] <= C_BOUND * table[k]["oracle32_p999"] + 2e-7, (k, table[k])


def test_small_gradient_tensors_error_is_the_references_own_over_16_seeds(dev):
    """depth_scale / depth_shift (the two tensors optimizer_ss steps on, :954 / :996) and the other
    few-element gradients are signed sums over (ray, sample, argmin-hypothesis) events that largely cancel:
    the reference's OWN fp32 arithmetic is 1e-4 ... 1e-1 away from the fp64 answer on them, differently on
    every draw.  So the bound is a statistic over 16 seeded problems (rays, targets, hypotheses, draws, both
    networks re-drawn): median over seeds of err(HIP) / err(reference fp32) <= 2, and the median errors
    themselves within the same factor - no absolute floor."""
    seeds = [1000 + 17 * i for i in range(16)]
    per_key = {}
    for sd in seeds:
        hipg, g32, g64 = _train_step_grads(dev, 128, 20, seed=sd)
        for k in g64:
            if k == "loss" or not _is_small(g64[k]) or float(g64[k].abs().max()) == 0.0:
                continue
            ref = g64[k].detach().double().cpu().flatten()
            den = float(ref.norm()) + 1e-300
            eh = float((hipg[k].detach().double().cpu().flatten() - ref).norm()) / den
            eo = float((g32[k].detach().double().cpu().flatten() - ref).norm()) / den
            per_key.setdefault(k, []).append((eh, eo))
    med = lambda v: float(torch.tensor(v, dtype=torch.float64).median())
    table, bad = {}, []
    for k, pairs in per_key.items():
        eh, eo = [p[0] for p in pairs], [p[1] for p in pairs]
        ratios = [h / max(o, 1e-300) for h, o in pairs]
        table[k] = {"seeds": len(pairs), "median_ratio_hip_over_oracle32": med(ratios),
                    "median_hip_rel_err": med(eh), "median_oracle32_rel_err": med(eo),
                    "max_hip_rel_err": max(eh), "max_oracle32_rel_err": max(eo), "ratios": ratios}
        if med(ratios) > C_BOUND or med(eh) > C_BOUND * med(eo) + 2e-7:
            bad.append((k, table[k]["median_ratio_hip_over_oracle32"], med(eh), med(eo)))
    _dump("small_gradient_tensors_16seeds_128rays_K20", table)
    assert {"depth_scale", "depth_shift"} <= set(table), sorted(table)
    assert not bad, "few-element gradients: HIP farther from fp64 than C x the reference's fp32 (medians over seeds): %s" % bad
