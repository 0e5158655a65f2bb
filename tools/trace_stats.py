#!/usr/bin/env python3
"""Per-dispatch rocprofv3 output -> statistics over the TIMED launches only, one row per launch size.

rocprofv3's own ``--stats`` average of a kernel holds every launch of the process - setup, warm-up and
clock-ramp launches included - and mixes the launch sizes.  This reads the per-dispatch files instead:

    trace_stats.py timed <kernel_trace.csv> <last_n> [substring ...]  > stats.csv
        per (kernel, grid size): the LAST ``last_n`` launches (the timed steps of bench.py; earlier ones
        are setup / warm-up) -> launches, average / min / max / stddev in ns

    trace_stats.py clock <counter_collection.csv> <kernel_trace.csv> <last_n> [substring ...]  > clock.csv
        per (kernel, grid size): effective shader clock of the last ``last_n`` launches =
        (GRBM_GUI_ACTIVE / 8 XCDs) / dispatch duration, plus SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over the
        same cycles (MFMA busy) when that counter is in the pass (MI355X_MICROARCH.md, DVFS give-back)

Runs on the GPU box (tools/profile_gpu.sh) and anywhere else (plain csv in, csv out).
"""
import collections
import csv
import math
import sys


def _grid(r):
    for k in ("Grid_Size_X", "Grid_Size"):
        if r.get(k) not in (None, ""):
            return int(float(r[k]))
    return 0


def _want(name, subs):
    return not subs or any(s in name for s in subs)


def timed(trace, last_n, subs):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if _want(r["Kernel_Name"], subs):
            rows[(r["Kernel_Name"], _grid(r))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Grid_Size", "Launches_In_Process", "Launches_Timed", "AverageNs", "MinNs", "MaxNs", "StdDevNs"])
    for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        v.sort()
        d = [e - s for s, e in v[-last_n:]]
        m = sum(d) / len(d)
        sd = math.sqrt(sum((x - m) ** 2 for x in d) / len(d))
        w.writerow([name, grid, len(v), len(d), f"{m:.1f}", min(d), max(d), f"{sd:.1f}"])


def clock(counters, trace, last_n, subs):
    dur = {}
    for r in csv.DictReader(open(trace)):
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    per = collections.defaultdict(lambda: collections.defaultdict(dict))     # (kernel, grid) -> dispatch -> counter -> value
    for r in csv.DictReader(open(counters)):
        if _want(r["Kernel_Name"], subs):
            per[(r["Kernel_Name"], _grid(r))][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Grid_Size", "Launches_Timed", "AverageNs", "GRBM_GUI_ACTIVE_mean", "Effective_Clock_GHz",
                "MFMA_Busy_Frac"])
    for (name, grid), disp in sorted(per.items()):
        ids = sorted((d for d in disp if d in dur), key=lambda d: dur[d][0])[-last_n:]
        if not ids or "GRBM_GUI_ACTIVE" not in disp[ids[0]]:
            continue
        ns = [dur[d][1] - dur[d][0] for d in ids]
        gui = [disp[d]["GRBM_GUI_ACTIVE"] for d in ids]
        ghz = [g / 8.0 / n for g, n in zip(gui, ns)]
        busy = [disp[d]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (g / 8.0) for d, g in zip(ids, gui)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in disp[d] and g > 0]
        w.writerow([name, grid, len(ids), f"{sum(ns) / len(ns):.1f}", f"{sum(gui) / len(gui):.0f}",
                    f"{sum(ghz) / len(ghz):.4f}", f"{sum(busy) / len(busy):.4f}" if busy else ""])


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "timed":
        timed(sys.argv[2], int(sys.argv[3]), sys.argv[4:])
    elif len(sys.argv) >= 5 and sys.argv[1] == "clock":
        clock(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5:])
    else:
        sys.exit(__doc__)
