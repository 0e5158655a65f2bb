#!/usr/bin/env python3
"""Regenerates the status table of DESIGN.md section 5 (between <!-- BEGIN status --> / <!-- END status -->) from the
committed evidence, so that its counts and figures cannot go stale:

    profiles/<tag>_bench_line.json            the un-profiled default `python bench.py` line of the round
    profiles/<tag>_bench_line_4096rays_k40.json   the BASELINE config-5 shapes on one GPU (optional)
    profiles/<tag>_gputests.txt               tail of `python -m pytest tests -m gpu -q` on the MI355X
    (CPU test count: `python -m pytest tests -m "not gpu" --collect-only -q` run here)

Usage: python tools/design_status.py r05"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")
d = json.loads(open(P("bench_line.json")).read().strip().splitlines()[-1])
big = json.loads(open(P("bench_line_4096rays_k40.json")).read().strip().splitlines()[-1]) if os.path.exists(P("bench_line_4096rays_k40.json")) else None
gpu = open(P("gputests.txt")).read() if os.path.exists(P("gputests.txt")) else ""
m = re.search(r"(\d+) passed", gpu)
n_gpu = m.group(1) if m else "?"
cpu = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "not gpu", "--collect-only", "-q"], cwd=ROOT,
                     capture_output=True, text=True).stdout
m = re.search(r"(\d+)/\d+ tests collected|(\d+) tests? collected", cpu)
n_cpu = (m.group(1) or m.group(2)) if m else "?"


def ts(key, src=d):
    r = src.get(key)
    if not isinstance(r, dict) or "ms_per_step" not in r:
        return None
    b = r.get("ms_per_step_blocks")
    spread = f" (blocks {b['min']:.3f}–{b['max']:.3f})" if b else ""
    return f"{r['ms_per_step']:.3f} ms{spread} ⇒ {r['value'] / 1e3:.0f} k rays/s, {r['whole_step_frac_of_peak']:.3f} of {r['peak']}"


rows = []
rf = d["roofline"]
rows.append(("headline: `python bench.py` (BASELINE `configs[1]` shape, synthetic rays / weights: test-render step, 1024 rays × (64+128), fp32)",
             f"**{d['value'] / 1e3:.1f} k rays/s, {d['ms_per_step']:.4f} ms/step**", "`value`, `ms_per_step`"))
rows.append(("roofline of the dominant kernel (`mlp_fwd_kernel`, HIP events around its launches)",
             f"**{rf['achieved']:.1f} TFLOP/s = {rf['frac']:.4f}** of the {rf['peak']} TFLOP/s fp32-MFMA peak; {rf['traffic'] / 1e6:.1f} MB HBM per launch "
             f"(static, from `profiles/`), MFMA pipe {100 * rf.get('mfma_util_pmc', 0):.1f} % busy", "`roofline`"))
cb = d.get("cpu_baseline")
if cb:
    tr = cb.get("train_step", {})
    rows.append((f"CPU baseline (oracle = bit-exact restatement of the reference, {cb['cores']} host cores, best of 5)",
                 f"forward {cb['value']:.0f} rays/s ⇒ GPU/CPU {d.get('gpu_over_cpu', 0):.0f}×" +
                 (f"; train step {tr.get('value', 0):.0f} rays/s ⇒ {d.get('gpu_over_cpu_train_step', 0):.0f}×" if tr else ""),
                 "`cpu_baseline`"))
for key, name in (("train_step", "train step, exact fp32 (`configs[2]`: fwd + 3-term loss with K = 20 + bwd + fused Adam)"),
                  ("train_step_f16x3", "train step, split precision f16x3 (fp32-class error)"),
                  ("train_step_bf16", "train step, bf16 with 16-bit saved rows"),
                  ("train_step_bf16_s8", "train step, bf16 with 8-bit saved rows (format code 2)")):
    v = ts(key)
    if v:
        rows.append((name, v, f"`{key}`"))
for key, name in (("fast_path_f16x3", "render step, f16x3"), ("fast_path_bf16", "render step, bf16")):
    r = d.get(key)
    if isinstance(r, dict) and "ms_per_step" in r:
        rows.append((name, f"{r['ms_per_step']:.3f} ms ⇒ {r['value'] / 1e6:.2f} M rays/s (MLP {r['roofline']['frac']:.3f} of {float(r['roofline']['peak']):.0f} {r['roofline']['unit']})", f"`{key}`"))
g = d.get("train_step_graph")
if isinstance(g, list):
    cells = []
    for e in g:
        sp = e.get("spread_graph_ms")
        cells.append(f"{e['rays']} rays {e['precision']}: eager {e['ms_per_step_eager']:.3f} / graph **{e['ms_per_step_graph']:.3f}** ms" +
                     (f" [{sp[0]:.3f}–{sp[1]:.3f}]" if sp else ""))
    rows.append(("train step eager vs one HIP graph (interleaved A/B, medians of 7 blocks)", "; ".join(cells), "`train_step_graph`"))
dl = d.get("driver_loop")
if isinstance(dl, list):
    cells = [f"{e['rays']} rays {e['precision']}: **{e['ms_per_iteration']:.3f}** ms / iteration" for e in dl if isinstance(e, dict) and "ms_per_iteration" in e]
    rows.append(("the training LOOP (`driver.train_scene`: view pick, pixel pick, fused batch gather into the captured step's inputs, "
                 "graph replay; 468 × 624, 18 views, K = 20; steady state, host clock)", "; ".join(cells), "`driver_loop`"))
di = d.get("train_step_dropin")
di = [di] if isinstance(di, dict) else di
if isinstance(di, list) and di and all(isinstance(e, dict) and "ms_per_step" in e for e in di):
    rows.append(("the drop-in operator path of INTEGRATION.md §2 (public operators + `loss.backward()` + `torch.optim.Adam` × 2, eager)",
                 "; ".join(f"{e['rays']} rays {e.get('precision', 'f32')}: {e['ms_per_step']:.3f} ms / step" for e in di),
                 "`train_step_dropin`"))
ce = d.get("strong_scaling_ceiling_8gpu")
if isinstance(ce, dict) and ce:
    rows.append(("strong-scaling ceiling of a 1024-ray batch on 8 GPUs before any RCCL time = ms(1024-ray step) / ms(128-ray graphed shard)",
                 "; ".join(f"{k} {v:.2f}×" for k, v in ce.items()), "`strong_scaling_ceiling_8gpu`"))
if big:
    cells = [f"render {big['ms_per_step']:.3f} ms ({big['value'] / 1e3:.0f} k rays/s)"]
    for key in ("train_step", "train_step_f16x3", "train_step_bf16", "train_step_bf16_s8"):
        r = big.get(key)
        if isinstance(r, dict) and "ms_per_step" in r:
            cells.append(f"{key} {r['ms_per_step']:.3f} ms ({r['whole_step_frac_of_peak']:.3f})")
    rows.append(("config-5 shapes on one GPU (4096 rays, K = 40)", "; ".join(cells), f"`profiles/{tag}_bench_line_4096rays_k40.json`"))
rows.append(("parity", f"{n_gpu} GPU tests (`-m gpu`, MI355X) + {n_cpu} CPU tests (`-m \"not gpu\"`) green", f"`profiles/{tag}_gputests.txt`, `tests/`"))

out = ["| item | value | where |", "|---|---|---|"] + [f"| {a} | {b} | {c} |" for a, b, c in rows]
block = "<!-- BEGIN status -->\n" + "\n".join(out) + "\n<!-- END status -->"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
if "<!-- BEGIN status -->" in s:
    s = re.sub(r"<!-- BEGIN status -->.*?<!-- END status -->", lambda _: block, s, flags=re.S)
    open(path, "w").write(s)
    print("DESIGN.md status table regenerated")
else:
    print(block)
