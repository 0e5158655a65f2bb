#!/usr/bin/env python3
"""Interleaved A/B of two (or more) builds of libscade_hip.so on the train step: every variant is run as its own process
(the library is chosen at load time: SCADE_LIB), in ALTERNATING order, ``--reps`` times each; median and spread of
tools/probe_step.py's ms / step.  The harness behind every kernel claim below ~5 %.

    SCADE_AB_FLAGS=-DX SCADE_AB_OUT=tools/scratch/ab_X python -m scade_amd.build
    python tools/ab_step.py bf16-s8 1024 intree tools/scratch/ab_X/libscade_hip.so [--reps 5] [--graph]"""
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
if "--reps" in sys.argv:
    args.remove(str(reps))
prec, rays, libs = args[0], args[1], args[2:]
res = {l: [] for l in libs}
for _ in range(reps):
    for l in libs:
        env = dict(os.environ)
        if l != "intree":
            env["SCADE_LIB"] = os.path.abspath(l)
        else:
            env.pop("SCADE_LIB", None)
        cmd = [sys.executable, os.path.join(here, "probe_step.py"), prec, rays] + (["graph"] if "--graph" in sys.argv else [])
        o = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300).stdout
        m = re.search(r"([0-9.]+) ms / step", o)
        if m:
            res[l].append(float(m.group(1)))
for l, v in res.items():
    v = sorted(v)
    print(f"{l:48s} median {v[len(v) // 2]:.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}  ({len(v)} runs)")
