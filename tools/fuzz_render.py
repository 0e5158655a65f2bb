"""Randomized differential run of the inference path (render_rays, no_grad): for every seed a random configuration
(rays, coarse / fine sample counts, lindisp, jittered or deterministic draws, chunk length, inference precision)

  * exact (f32) render against the oracle: the coarse stage - identical inputs - element-wise to 1e-4; everything behind
    the ill-conditioned resampling by PSNR (> 60 dB) and norm (1e-3), as tests/test_gpu_render.py does;
  * chunked (batchify_rays at a random chunk length, one and two streams) == whole batch, bit for bit;
  * HIP-graph replay (GraphedRender) == eager, bit for bit (deterministic draws);
  * the fast inference precisions against the exact render: f16x3 PSNR > 70 dB, f16 / bf16 > 30 dB, finite.

  python tools/fuzz_render.py --seeds 200 [--first 0] [--out FILE]
"""
import argparse, json, os, sys, time, traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import scade_amd as S                                     # noqa: E402
from oracle import scade_oracle as O                      # noqa: E402  (test infrastructure: the checker)
from scade_amd.graphs import GraphedRender                # noqa: E402
from scade_amd.train import make_scade_nets               # noqa: E402

MAPS = ["rgb_map", "depth_map", "acc_map"]
FAST_PSNR = {"f16x3": 70.0, "f16": 30.0, "bf16": 30.0}
ALWAYS_FP64 = False

MAX_RAYS = 700             # --max-rays: larger batches (the 128-point tiles, several rounds of workgroups) without the oracle leg


def rel_l2(a, b):
    a, b = torch.nan_to_num(a.double().reshape(-1).cpu()), torch.nan_to_num(b.double().reshape(-1).cpu())
    return float((a - b).norm() / (b.norm() + 1e-300))


def psnr(a, b):
    return float(-10 * torch.log10(torch.mean((a.double().cpu() - b.double().cpu()) ** 2) + 1e-30))


def config(seed):
    g = torch.Generator().manual_seed(11000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    c = dict(seed=seed, N=ri(1, MAX_RAYS), Ns=ri(3, 130), Ni=ri(1, 200), lindisp=bool(ri(0, 3) == 0), jitter=bool(ri(0, 1)),
             fast=["f16x3", "f16", "bf16"][ri(0, 2)])
    if ri(0, 3) == 0:
        c["Ns"], c["Ni"] = 64, 128
    c["chunk"] = ri(1, max(1, c["N"]))
    return c, g


def same(a, b):
    return all(torch.equal(torch.nan_to_num(a[k]), torch.nan_to_num(b[k])) for k in a)


def one(c, g, dev, oracle):
    row = dict(c)
    N, ns, ni = c["N"], c["Ns"], c["Ni"]
    rays = O.synthetic_rays(N, seed=12000 + c["seed"])
    coarse, fine = make_scade_nets(dev, seed=1 + c["seed"] % 7)
    embed_fn, _ = S.get_embedder(9, 0)
    embeddirs_fn, _ = S.get_embedder(0, 0)
    bbc, bbs = torch.zeros(3), torch.tensor(0.2)
    query = S.make_network_query_fn(embed_fn, embeddirs_fn, bbc.to(dev), bbs.to(dev))
    draws = tuple(torch.rand(N, n, generator=g) for n in (ns, ni, ni)) if c["jitter"] else None
    kw = dict(N_importance=ni, network_fine=fine, lindisp=c["lindisp"], retraw=True)
    if draws:
        kw.update(perturb=1., t_rand=draws[0].to(dev), u_coarse=draws[1].to(dev), cached_u=draws[2].to(dev))
    else:
        kw.update(perturb=0.)
    rd = rays.to(dev)
    with torch.no_grad():
        ret = S.render_rays(rd, True, coarse, query, ns, **kw)
        # chunked == whole (one stream, two streams)
        bk = dict(network_fn=coarse, network_query_fn=query, N_samples=ns, **kw)
        if draws:                                         # (per-ray draws are sliced by the caller in the reference too)
            bk = None
        if bk is not None:
            row["chunked_bitwise"] = same(S.batchify_rays(rd, c["chunk"], True, **bk), ret)
            row["chunked_2streams_bitwise"] = same(S.batchify_rays(rd, c["chunk"], True, streams=2, **bk), ret)
            gr = GraphedRender(N, coarse, query, ns, ni, fine, lindisp=c["lindisp"], retraw=True)
            a = {k: v.clone() for k, v in gr(rd).items()}
            b = {k: v.clone() for k, v in gr(rd).items()}            # (second call: a replay without capture)
            row["graph_bitwise"] = same(a, ret) and same(b, ret)
        # fast inference precisions against the exact render
        coarse.inference_precision = fine.inference_precision = c["fast"]
        fast = S.render_rays(rd, True, coarse, query, ns, **kw)
        coarse.inference_precision = fine.inference_precision = "f32"
        row["fast_psnr"] = psnr(fast["rgb_map"], ret["rgb_map"])
        row["fast_finite"] = bool(all(torch.isfinite(fast[k]).all() for k in MAPS))
    z = ret["z_vals"]
    row["sorted"] = bool((z[:, 1:] >= z[:, :-1]).all())
    row["shapes"] = tuple(z.shape) == (N, ns + ni) and tuple(ret["pred_hyp"].shape) == (N, ni)
    ok = row["sorted"] and row["shapes"] and row["fast_finite"] and row["fast_psnr"] > FAST_PSNR[c["fast"]] \
        and row.get("chunked_bitwise", True) and row.get("chunked_2streams_bitwise", True) and row.get("graph_bitwise", True)
    if oracle and N * (ns + ni) <= 60000:
        pc = {k: v.detach().cpu() for k, v in coarse.named_parameters()}
        pf = {k: v.detach().cpu() for k, v in fine.named_parameters()}
        with torch.no_grad():
            want = O.render_rays(rays, pc, pf, bbc, bbs, n_samples=ns, n_importance=ni, lindisp=c["lindisp"], retraw=True,
                                 **(dict(t_rand=draws[0], u_coarse=draws[1], u_fine=draws[2]) if draws else {}))
        worst = 0.0
        for k in ("rgb0", "depth0", "weights0", "z_vals0"):
            w = want[k]
            err = (ret[k].cpu() - w).abs() - (1e-4 * w.abs() + 2e-6)
            worst = max(worst, float(err.max()))
        row["coarse_elementwise_excess"] = worst                     # <= 0: inside rtol 1e-4 / atol 2e-6
        row["rgb_psnr"] = psnr(ret["rgb_map"], want["rgb_map"])
        row["rgb_rel_l2"], row["depth_rel_l2"] = rel_l2(ret["rgb_map"], want["rgb_map"]), rel_l2(ret["depth_map"], want["depth_map"])
        fine_ok = row["rgb_psnr"] > 60 and row["rgb_rel_l2"] < 1e-3 and row["depth_rel_l2"] < 1e-3
        # per ray: the norm is carried by the few rays where a sample changed its cdf bin
        dr = ((ret["depth_map"].cpu() - want["depth_map"]).abs() / (want["depth_map"].abs() + 1e-6)).double()
        row["depth_per_ray"] = dict(median=float(dr.median()), p95=float(dr.quantile(0.95)), max=float(dr.max()),
                                    over_1e3=float((dr > 1e-3).double().mean()))
        row["z_vals0_bitwise"] = bool(torch.equal(ret["z_vals0"].cpu(), want["z_vals0"]))
        if not fine_ok or ALWAYS_FP64:
            # what can two correct fp32 evaluations differ by HERE?  The distance of the reference's fp32 arithmetic to an
            # fp64 evaluation of the same algorithm is the yardstick (tests/test_gpu_parity64.py): the kernels must be as
            # close to the fp64 result as torch's fp32 is, within a factor of two
            prev = torch.get_default_dtype()
            torch.set_default_dtype(torch.float64)
            try:
                d64 = lambda t: t.double()
                with torch.no_grad():
                    w64 = O.render_rays(d64(rays), {k: d64(v) for k, v in pc.items()}, {k: d64(v) for k, v in pf.items()},
                                        d64(bbc), d64(bbs), n_samples=ns, n_importance=ni, lindisp=c["lindisp"],
                                        **(dict(t_rand=d64(draws[0]), u_coarse=d64(draws[1]), u_fine=d64(draws[2])) if draws else {}))
            finally:
                torch.set_default_dtype(prev)
            row["vs_fp64"] = {k: [rel_l2(ret[k], w64[k]), rel_l2(want[k], w64[k])] for k in ("rgb_map", "depth_map")}
            # per ray: a sample that changes its cdf bin moves ONE ray's depth by percents and carries the norm (seed 370:
            # one ray of 222 at 3.4e-2, median 3.8e-6); medians and 95th percentiles are what an arithmetic deficit moves
            def per_ray(x, y, rel):
                e = (x.double().cpu() - y.double()).abs()
                e = e / (y.double().abs() + 1e-6) if rel else e.amax(-1)
                return dict(median=float(e.median()), p95=float(e.quantile(0.95)), max=float(e.max()))
            row["per_ray_vs_fp64"] = {k: dict(kernels=per_ray(ret[k], w64[k], k == "depth_map"), torch_fp32=per_ray(want[k], w64[k], k == "depth_map"))
                                      for k in ("rgb_map", "depth_map")}
            # (factor 2.5: the samples that DO change bins come in clusters - a ray's deterministic draws share the bin -
            # and their count differs between two fp32 evaluations: seed 1076, 87 against 34 of 46,230 samples off by more
            # than 1e-3 while the rest agree with fp64 to 2.5e-5 / 2.0e-5, and the sampler alone, on identical fp32 inputs,
            # is ten times CLOSER to fp64 than torch's: 4.4e-7 against 4.7e-6)
            # (under 100 rays the 95th percentile is the second or third worst ray: the median alone)
            robust = all(v["kernels"]["median"] <= 2.5 * v["torch_fp32"]["median"] + 1e-6
                         and (N < 100 or v["kernels"]["p95"] <= 2.5 * v["torch_fp32"]["p95"] + 1e-5)
                         for v in row["per_ray_vs_fp64"].values())
            fine_ok = fine_ok or all(a <= 2 * b + 1e-5 for a, b in row["vs_fp64"].values()) or (robust and row["rgb_psnr"] > 55)
        ok = ok and worst <= 0 and fine_ok
    row["ok"] = bool(ok)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-rays", type=int, default=700)
    ap.add_argument("--fp64", action="store_true", help="evaluate the oracle in fp64 for every row (statistics)")
    ap.add_argument("--only-lindisp", action="store_true")
    a = ap.parse_args()
    globals()["MAX_RAYS"] = a.max_rays
    global ALWAYS_FP64
    ALWAYS_FP64 = a.fp64
    dev = torch.device("cuda", 0)
    rows, bad, t0 = [], 0, time.time()
    for seed in range(a.first, a.first + a.seeds):
        c, g = config(seed)
        if a.only_lindisp and not c["lindisp"]:
            continue
        try:
            row = one(c, g, dev, not a.no_oracle)
        except Exception as e:
            row = dict(c, ok=False, error="%s: %s" % (type(e).__name__, str(e)[:300]))
            traceback.print_exc()
        rows.append(row)
        bad += not row["ok"]
        print(json.dumps(row), flush=True)
    print("fuzz_render: %d configurations, %d failed, %.0f s" % (len(rows), bad, time.time() - t0))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(configurations=len(rows), failed=bad, rows=rows), f, indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
