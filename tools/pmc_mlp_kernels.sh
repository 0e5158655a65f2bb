# Run on the GPU box: PMC breakdown (MFMA busy, VALU/LDS/VMEM activity, waits, L2 hits) of the MLP forward kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --no-cpu-baseline --no-image --no-train --steps 10 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/p1 -o b -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/p2 -o b -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/p3 -o b -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/p[123]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp_fwd" in k:
            rows[k[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: f"{x:.4g}" for c, x in sorted(m.items())})
    if "GRBM_GUI_ACTIVE" in m:
        print("   mfma_util", (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (m["GRBM_GUI_ACTIVE"] / 8),
              " wave_cycles/gui(quad)", m["SQ_WAVE_CYCLES"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8) / 2048,
              " wait_any frac", m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], " wait_inst frac", m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
              " valu frac", m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], "lds frac", m["SQ_ACTIVE_INST_LDS"] / m["SQ_WAVE_CYCLES"], "vmem frac", m["SQ_ACTIVE_INST_VMEM"] / m["SQ_WAVE_CYCLES"])
PY
