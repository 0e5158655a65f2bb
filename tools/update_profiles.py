#!/usr/bin/env python3
"""Copy the outputs of tools/profile_gpu.sh <tag> (gpurun_out/prof_<tag>/) and an un-profiled bench line
into profiles/ and (re)generate that round's reading block in profiles/README.md (between the markers
<!-- BEGIN <tag> --> / <!-- END <tag> -->; blocks of other rounds are left alone).
Refuses (non-zero exit, nothing copied) when the kernel statistics do not fit the bench lines they are meant to
back: the headline kernel's timed average of the coarse launch + that of the fine launch (= the MLP time of ONE
step) must not exceed ms_per_step of the profiled run's own line NOR of the un-profiled line of the same box,
and the rocprofv3-derived roofline fraction must agree with the same-run HIP-event fraction within 1 %.
Usage: python tools/update_profiles.py r03 gpurun_out/bench_final.json [--force]"""
import csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench = sys.argv[1], sys.argv[2]
force = "--force" in sys.argv
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles")
PEAK, FLOP_PER_POINT = 157.3, 2 * 587264


def load_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{"metric')][-1])


# ---- consistency gate -----------------------------------------------------------------
timed = [r for r in csv.DictReader(open(os.path.join(src, "kernel_stats_timed.csv")))
         if "mlp_fwd_kernel<1, false, 2>" in r["Name"]]
assert len(timed) == 2, f"expected the coarse and the fine launch size of the headline kernel, got {len(timed)} rows"
per_step_ms = sum(float(r["AverageNs"]) for r in timed) / 1e6
fl = sum(int(r["Grid_Size"]) // 4 * FLOP_PER_POINT for r in timed) / 2          # mean FLOP per launch
frac_prof = fl / (per_step_ms / 2 * 1e-3) / 1e12 / PEAK
under = load_line(os.path.join(src, "bench_line_under_rocprof.json"))
same_box = load_line(os.path.join(src, "bench_line_unprofiled_same_box.json"))
problems = []
for what, ln in (("the profiled run's own line", under), ("the un-profiled line of the same box", same_box)):
    if per_step_ms > ln["ms_per_step"]:
        problems.append(f"coarse + fine timed average {per_step_ms:.4f} ms exceeds ms_per_step {ln['ms_per_step']:.4f} of {what}")
if abs(frac_prof / under["roofline"]["frac"] - 1) > 0.01:
    problems.append(f"rocprofv3-derived frac {frac_prof:.4f} vs same-run HIP-event frac {under['roofline']['frac']:.4f}: > 1 %")
print(f"gate: rocprofv3 timed launches {per_step_ms:.4f} ms per step -> frac {frac_prof:.4f}; same-run HIP events "
      f"{under['roofline']['frac']:.4f}; un-profiled same box {same_box['roofline']['frac']:.4f} "
      f"(ms_per_step {same_box['ms_per_step']:.4f})")
if problems and not force:
    sys.exit("update_profiles: REFUSING to copy these profiles:\n  " + "\n  ".join(problems))

for a, b in (("kernel_stats_timed.csv", "render_kernel_stats_timed.csv"), ("render_clock.csv", "render_clock.csv"),
             ("train_clock.csv", "train_clock.csv"),
             ("bench_line_unprofiled_same_box.json", "render_bench_line_unprofiled_same_box.json"),
             ("kernel_stats.csv", "render_kernel_stats.csv"), ("kernel_stats_train.csv", "train_kernel_stats.csv"),
             ("bench_line_under_rocprof.json", "render_bench_line_under_rocprof.json"),
             ("bench_line_train_under_rocprof.json", "train_bench_line_under_rocprof.json"),
             ("pmc_summary.csv", "pmc_summary.csv"), ("pmc.json", "pmc.json")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))
line = [l for l in open(os.path.join(ROOT, bench)).read().splitlines() if l.startswith('{"metric')][-1]
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")

pm = json.load(open(os.path.join(dst, f"{tag}_pmc.json")))
d = json.loads(line)
dr = json.load(open(os.path.join(dst, f"{tag}_render_bench_line_under_rocprof.json")))
st = {r["Name"]: r for r in csv.DictReader(open(os.path.join(dst, f"{tag}_render_kernel_stats.csv")))}
fw = next(v for k, v in st.items() if "mlp_fwd_kernel<1, false, 2>" in k)
avg = per_step_ms / 2          # mean over the TIMED launches (coarse and fine in equal numbers)
fw = dict(fw, Calls=str(sum(int(r["Launches_Timed"]) for r in timed)))
size_rows = "\n".join(
    f"| {int(r['Grid_Size']) // 4:,} | {r['Launches_Timed']} of {r['Launches_In_Process']} | {float(r['AverageNs']) / 1e6:.4f} | "
    f"{int(r['MinNs']) / 1e6:.4f} | {int(r['MaxNs']) / 1e6:.4f} | "
    f"{int(r['Grid_Size']) // 4 * FLOP_PER_POINT / float(r['AverageNs']) / 1e3:.1f} | "
    f"{int(r['Grid_Size']) // 4 * FLOP_PER_POINT / float(r['AverageNs']) / 1e3 / PEAK:.4f} | "
    f"{pm['mlp_fwd_kernel'].get('timed_launches_by_points', {}).get(str(int(r['Grid_Size']) // 4), {}).get('effective_clock_ghz', float('nan')):.3f} |"
    for r in sorted(timed, key=lambda r: int(r["Grid_Size"])))
notes = {
    "mlp_fwd_kernel": f"headline render kernel; algorithmic HBM ~ 6.1 MB, the 2.4 MB weight blob is fetched once per XCD L2 "
                      f"(8 x 2.4 MB = the 4x over-fetch; 22 GB/s, harmless); rocprofv3 average {avg:.4f} ms over {fw['Calls']} "
                      f"launches vs HIP events {dr['roofline']['avg_launch_ms']:.4f} ms over the timed ones in the same run; "
                      f"{dr['roofline']['frac']:.3f} of the 157.3 TFLOP/s peak by timing",
    "mlp_fwd_kernel_train": "exact training forward (fp32 activation rows + one 32-bit ReLU word per lane, layer and point tile written)",
    "mlp_dgrad_kernel": "dZ rows written; round 3: one launch walks the tiles of BOTH networks of a step",
    "mlp_fwd_kernel_train_small": "32-point workgroups (the 128-ray launches of the graph region)",
    "mlp_dgrad_kernel_small": "32-point workgroups",
    "mlp_wgrad_kernel": "dZ and activations streamed once per layer; includes the low-MFMA embedding and rgb-head jobs",
    "mlp_wgrad2_kernel": "exact weight gradient: half-layer workgroups, two per CU, point-major tiles by LDS-DMA into a 3-slot ring, output rows permuted (no transposition); "
                         "round 3: ONE launch covers both networks of a step; the mean includes the 128-ray launches",
    "wgrad_reduce4_kernel": "sums the per-chunk partials",
    "mlp_fwd_f16_kernel": "opt-in f16x3 render kernel (3 f16 MFMAs per fp32-class product)",
    "mlp_fwd_f16_kernel_train": "f16x3 training forward; round 5: the saved rows leave as the fp16 h plane + an e5m2 l plane (768 of 1024 bytes per point and layer)",
    "mlp_dgrad_f16_kernel": "round 5: dZ rows as h + l8 planes in the per-point scaled domain, 1 / s_p per point beside them",
    "mlp_wgrad_f16_kernel": "round 5: contracts the h + l8 rows as they lie (v_perm_b32 pairs, no fp16 split); compute-bound now: memory-only knock-out 0.58 of the kernel",
    "mlp_fwd_lp_kernel_bf16": "opt-in bf16 render kernel (config 5); round 2: the epilogue trades chunk halves between lane r and "
                              "r + 32 (v_permlane32_swap) and stores conflict-free ds_write_b128 (was 26.4 % conflict cycles)",
    "mlp_fwd_lp_kernel_f16": "same kernel, fp16 operands",
    "mlp_fwd_lp_kernel_bf16_train": "bf16 training forward: 16-bit activation rows; round 3: conflict-free lane map of the tile copies",
    "mlp_dgrad_lp_kernel_bf16": "16-bit dZ rows, both networks in one launch; round 3: epilogue halves traded with v_permlane32_swap + ds_write_b128 (was 30.1 % conflict cycles)",
    "mlp_fwd_lp_kernel_bf16_s8_train": "format code 2: the bf16 training forward saving 8-bit e5m2 rows (round 4: as whole-row riders of the next k-loop), fp8 embedding rows, sign words for the views layer too, heads on the MFMA",
    "mlp_dgrad_lp_kernel_bf16_s8": "format code 2: 8-bit e5m2 dZ rows under the launch-wide loss scale (round 4: riders; heads as one MFMA per point tile, masked by sign words - the 16-bit views rows are no longer read: -200 MB)",
    "mlp_wgrad_lp_kernel_bf16_s8": "format code 2 (round 4): e5m2 rows by LDS-DMA into a ring, ds_read_b64_tr_b8 fragments, MX-scaled bf8 MFMA at twice the 16-bit rate; balanced persistent launch; 1.53 GB at the memory system's rate",
    "wgrad2_reduce_pair_kernel": "sums the per-chunk partials of both networks",
    "wgrad_lp_reduce_pair_kernel": "sums the per-chunk partials of both networks (16-bit path)",
    "wgrad_lp_reduce_kernel": "round 3: sums, per parameter, the partial rows of the job that owns it (balanced launch: one row per segment)",
    "stage_inputs_kernel": "round 3: the inputs of a graph-captured step copied into its static buffers in one launch",
    "mlp_pack_step_f32": "round 3: the four weight packs of a step (both networks, forward + transposed layout) in one launch",
    "mlp_pack_step_bf16": "the same for the 16-bit kernels (incl. the NaN census of the fp32 parameters)",
    "adam_step2_kernel": "round 3: both optimizers (networks; depth scale / shift) in one update launch",
    "mlp_fwd_lp_kernel_bf16_train_small": "64-point workgroups (128-ray launches)",
    "mlp_dgrad_lp_kernel_bf16_small": "64-point workgroups",
    "mlp_wgrad_lp_kernel_bf16": "HBM-bound (11.7 KB read per point over the 13 jobs; ~5.6 TB/s); round 3: balanced persistent launch, one workgroup per CU with host-planned stage ranges",
    "ray_tail_coarse": "per-ray work between the coarse and the fine MLP launch (composite, sampler, sort-merge, points)",
    "ray_tail_fine": "fine composite + depth-hypothesis sampler + z_std",
    "ray_tail_bwd_fine": "round 2: backward of the fine tail in one launch (sampler backward + compositing backward)",
    "train_loss_fwd": "round 2: the three-term train loss as one kernel (+ a one-wave reduce)",
    "train_loss_bwd": "its backward (+ a one-wave scale/shift reduce)",
    "ray_tail_train": "round 3: fine tail + three-term loss (forward and backward) + the backward of both tails in ONE launch; round 5: one ray per workgroup of FOUR waves (critical chain | coarse ray + loss value | scale / shift scatter | z_std)",
    "train_loss_fb_reduce": "the loss's one-workgroup reduce (loss terms, depth scale / shift gradient rows)",
}
rows = []
for k in notes:
    if k not in pm:
        continue
    v = pm[k]
    hb = v["hbm_bytes_per_launch"]
    hbs = f"{hb / 1e9:.2f} GB" if hb >= 1e9 else f"{hb / 1e6:.1f} MB"
    rows.append(f"| `{k}` | {100 * v['mfma_util']:.1f} % | {hbs} | {100 * v['lds_bank_conflict_frac']:.1f} % | {notes[k]} |")
tr = d.get("train_step", {})
block = f"""<!-- BEGIN {tag} -->
## Round {int(tag[1:])} reading (`{tag}_pmc.json`, `{tag}_*_kernel_stats.csv`)

Every figure is a MEAN PER LAUNCH over all launches of that kernel in the profiled command - the coarse
(65,536-point) and fine (196,608-point) pass of every 1024-ray step and, for the training kernels, the
128-ray launches of the graph region - so the byte counts are not those of one particular launch size.

| kernel | MFMA busy | HBM bytes / launch | LDS conflict cycles | note |
|---|---|---|---|---|
""" + "\n".join(rows) + f"""

Headline kernel `mlp_fwd_kernel<1,false,2>`, TIMED launches only (`{tag}_render_kernel_stats_timed.csv`: the last 100
launches of each size from the per-dispatch trace; `{tag}_render_kernel_stats.csv` is rocprofv3's own summary over all
launches of the process, setup and warm-up included), one row per launch size, clock from the separate PMC pass
(`{tag}_render_clock.csv`):

| points | launches (timed of all) | average ms | min | max | TFLOP/s | of 157.3 | effective clock GHz |
|---|---|---|---|---|---|---|---|
{size_rows}

Mean over both sizes **{avg:.4f} ms** -> **{frac_prof:.4f}** of the peak from the rocprofv3 trace; bench.py's own HIP-event
figure in the SAME profiled run (`{tag}_render_bench_line_under_rocprof.json`): {dr['roofline']['avg_launch_ms']:.4f} ms, frac
{dr['roofline']['frac']:.4f} (agree within {abs(avg / dr['roofline']['avg_launch_ms'] - 1) * 100:.2f} %); the same command UN-profiled on the
same box right before (`{tag}_render_bench_line_unprofiled_same_box.json`): frac {same_box['roofline']['frac']:.4f},
{same_box['ms_per_step']:.4f} ms per step (profiler effect {100 * (under['ms_per_step'] / same_box['ms_per_step'] - 1):+.2f} % on the step).

Un-profiled bench line of the same build (`{tag}_bench_line.json`): {d['value']:.0f} rays/s, {d['ms_per_step']:.3f} ms/step,
`roofline.achieved` {d['roofline']['achieved']:.1f} TFLOP/s (frac {d['roofline']['frac']:.4f}), exact train step {tr.get('ms_per_step', float('nan')):.3f} ms,
CPU baseline {d['cpu_baseline']['value']:.0f} rays/s forward / {d['cpu_baseline'].get('train_step', {}).get('value', float('nan')):.0f} rays/s train step on {d['cpu_baseline']['cores']} cores.
<!-- END {tag} -->
"""
p = os.path.join(dst, "README.md")
s = open(p).read()
b0, b1 = f"<!-- BEGIN {tag} -->", f"<!-- END {tag} -->\n"
if b0 in s:
    s = s[:s.index(b0)] + block + s[s.index(b1) + len(b1):]
else:
    s = s.rstrip("\n") + "\n\n" + block
open(p, "w").write(s)
print(block[-900:])
