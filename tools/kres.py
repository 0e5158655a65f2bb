#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kres.py mlp_bwd_lp [substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scade_amd.build import CFLAGS, hipcc
src = os.path.join(ROOT, "scade_amd", "csrc", sys.argv[1] + ".hip")
sub = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run([hipcc()] + CFLAGS + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    if sub in k:
        print(f"{k[:95]:95s} vgpr {v.get('VGPRs','?'):>4s} agpr {v.get('AGPRs','?'):>3s} spill {v.get('VGPR Spill','?'):>3s} "
              f"scratch {v.get('ScratchSize [bytes/lane]','?'):>4s} sgpr {v.get('SGPRs','?'):>3s} occ {v.get('Occupancy [waves/SIMD]','?')} "
              f"lds {v.get('LDS Size [bytes/block]','?')}")
