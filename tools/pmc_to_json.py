#!/usr/bin/env python3
"""pmc_summary.csv (tools/profile_gpu.sh) -> profiles/rNN_pmc.json: per-kernel means per dispatch.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB  (the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md, HBM / rocprofv3 section); MFMA utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024
SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); LDS conflict fraction = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).  Optional further arguments: the
``clock`` tables of tools/trace_stats.py (train command: every launch; render command: the timed launches of
the headline kernel, one row per launch size) -> ``effective_clock_ghz`` = GRBM_GUI_ACTIVE / 8 XCDs /
dispatch duration per kernel (the 157.3 TFLOP/s and 2.5 PFLOP/s peaks assume 2.4 GHz)."""
import collections, csv, json, re, sys

KEYS = [  # (regex on the kernel name, key in the json); "_small" = the half-size workgroup variants
    (r"mlp_fwd_kernel<1, false, 2>", "mlp_fwd_kernel"), (r"mlp_fwd_kernel<1, true, 2>", "mlp_fwd_kernel_train"),
    (r"mlp_fwd_kernel<1, false, 1>", "mlp_fwd_kernel_small"), (r"mlp_fwd_kernel<1, true, 1>", "mlp_fwd_kernel_train_small"),
    (r"mlp_dgrad_kernel<2>", "mlp_dgrad_kernel"), (r"mlp_dgrad_kernel<1>", "mlp_dgrad_kernel_small"),
    (r"mlp_wgrad_kernel", "mlp_wgrad_kernel"), (r"mlp_wgrad2_kernel", "mlp_wgrad2_kernel"),
    (r"wgrad2_reduce_kernel", "wgrad2_reduce_kernel"), (r"wgrad2_rgb_kernel", "wgrad2_rgb_kernel"),
    (r"ray_tail_kernel<1, 4>", "ray_tail_coarse"), (r"ray_tail_kernel0<3>", "ray_tail_fine"),
    (r"ray_tail_bwd_kernel<3>", "ray_tail_bwd_fine"), (r"train_loss_fwd_kernel", "train_loss_fwd"),
    (r"train_loss_bwd_kernel", "train_loss_bwd"), (r"ray_tail_train_kernel<3>", "ray_tail_train"),
    # (round 5: one ray per workgroup of four waves up to four waves per SIMD, else the one-wave-per-ray form)
    (r"ray_tail_train_split<3>", "ray_tail_train"), (r"ray_tail_train_seq<3>", "ray_tail_train_seq"),
    (r"train_loss_fb_reduce_kernel", "train_loss_fb_reduce"),
    (r"mlp_fwd_f16_kernel<1, (false|0)>", "mlp_fwd_f16_kernel"), (r"mlp_fwd_f16_kernel<1, (true|1|2)>", "mlp_fwd_f16_kernel_train"),
    (r"mlp_dgrad_f16_kernel", "mlp_dgrad_f16_kernel"), (r"mlp_wgrad_f16_kernel", "mlp_wgrad_f16_kernel"),
    # (round 3: the SAVE template argument of the 16-bit forward is an int - 0 inference, 1 16-bit rows, 2 8-bit rows;
    # dgrad / wgrad carry an S8 flag; the backward kernels of a train step cover both networks in one launch)
    (r"mlp_fwd_lp_kernel<true, 1, (false|0), 4>", "mlp_fwd_lp_kernel_bf16"), (r"mlp_fwd_lp_kernel<false, 1, (false|0), 4>", "mlp_fwd_lp_kernel_f16"),
    (r"mlp_fwd_lp_kernel<true, 1, (true|1), 4>", "mlp_fwd_lp_kernel_bf16_train"),
    (r"mlp_fwd_lp_kernel<true, 1, (true|1), 2>", "mlp_fwd_lp_kernel_bf16_train_small"),
    (r"mlp_fwd_lp_kernel<true, 1, 2, 4>", "mlp_fwd_lp_kernel_bf16_s8_train"),
    (r"mlp_fwd_lp_kernel<true, 1, 2, 2>", "mlp_fwd_lp_kernel_bf16_s8_train_small"),
    (r"mlp_dgrad_lp_kernel<true, 4(, false)?>", "mlp_dgrad_lp_kernel_bf16"), (r"mlp_dgrad_lp_kernel<true, 2(, false)?>", "mlp_dgrad_lp_kernel_bf16_small"),
    (r"mlp_dgrad_lp_kernel<true, 4, true>", "mlp_dgrad_lp_kernel_bf16_s8"), (r"mlp_dgrad_lp_kernel<true, 2, true>", "mlp_dgrad_lp_kernel_bf16_s8_small"),
    (r"mlp_wgrad_lp_kernel<true(, false)?>", "mlp_wgrad_lp_kernel_bf16"), (r"mlp_wgrad_lp_kernel<true, true>", "mlp_wgrad_lp_kernel_bf16_s8"),
    (r"wgrad_reduce4_kernel", "wgrad_reduce4_kernel"), (r"wgrad2_reduce_pair_kernel", "wgrad2_reduce_pair_kernel"),
    (r"wgrad_lp_reduce_pair_kernel", "wgrad_lp_reduce_pair_kernel"), (r"wgrad_lp_reduce_kernel", "wgrad_lp_reduce_kernel"),
    (r"stage_inputs_kernel", "stage_inputs_kernel"),
    (r"mlp_pack_step_kernel<0>", "mlp_pack_step_f32"), (r"mlp_pack_step_kernel<1>", "mlp_pack_step_bf16"),
    (r"adam_step2_kernel", "adam_step2_kernel"), (r"ray_points_kernel", "ray_points_kernel"),
]


def _key(name):
    return next((k for rx, k in KEYS if re.search(rx, name)), None)


def clocks(path):
    """clock csv -> {key: {"ghz", "launches", "by_grid": {grid: {...}}}} (launch-weighted mean over grid sizes)."""
    out = {}
    for r in csv.DictReader(open(path)):
        key = _key(r["Name"])
        if key is None:
            continue
        n, ghz = int(r["Launches_Timed"]), float(r["Effective_Clock_GHz"])
        d = out.setdefault(key, {"n": 0, "sum": 0.0, "by_grid": {}})
        d["n"] += n
        d["sum"] += n * ghz
        d["by_grid"][r["Grid_Size"]] = {"launches": n, "avg_ms": float(r["AverageNs"]) / 1e6, "effective_clock_ghz": ghz,
                                        "mfma_busy": float(r["MFMA_Busy_Frac"]) if r["MFMA_Busy_Frac"] else None}
    return {k: {"ghz": d["sum"] / d["n"], "launches": d["n"], "by_grid": d["by_grid"]} for k, d in out.items()}


def main(src, dst, train_clock=None, render_clock=None):
    rows = collections.defaultdict(dict)
    for r in csv.reader(open(src)):
        if r[0] == "kernel":
            continue
        rows[r[0]][r[1]] = float(r[3])
    out = {}
    for name, v in rows.items():
        key = _key(name)
        if key is None:
            continue
        gui, mf = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        out[key] = {
            "fetch_size_kb": v.get("FETCH_SIZE"), "write_size_kb": v.get("WRITE_SIZE"),
            "hbm_bytes_per_launch": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0,
            "mfma_util": (mf / 1024) / (gui / 8) if gui else None,
            "lds_bank_conflict_frac": v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
            "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
            "wave_wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0),
        }
    if train_clock:
        for k, c in clocks(train_clock).items():
            if k in out:
                out[k]["effective_clock_ghz"] = c["ghz"]
    if render_clock:
        c = clocks(render_clock).get("mlp_fwd_kernel")
        if c and "mlp_fwd_kernel" in out:
            # the headline kernel: the TIMED launches of the render command, one row per launch size
            # (grid = threads: 256 per 64-point workgroup -> points = grid / 4)
            out["mlp_fwd_kernel"]["effective_clock_ghz"] = c["ghz"]
            out["mlp_fwd_kernel"]["timed_launches_by_points"] = {
                str(int(g) // 4): v for g, v in sorted(c["by_grid"].items(), key=lambda kv: int(kv[0]))}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:32s} mfma {v['mfma_util'] or 0:.3f}  hbm {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"lds-conflict {v['lds_bank_conflict_frac']:.3f}  L2 hit {v['l2_hit_rate'] or 0:.3f}  "
              f"clock {v.get('effective_clock_ghz') or 0:.3f} GHz")


if __name__ == "__main__":
    main(*sys.argv[1:5])
