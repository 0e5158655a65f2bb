# Run on the GPU box: PMC breakdown of the training kernels (tools/probe_train.py [precision])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/probe_train.py ${1:-f32}"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/q1 -o b -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/q2 -o b -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/q[12]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp_" in k and "pack" not in k:
            rows[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" not in m: continue
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{k:40s} mfma_util {(m['SQ_VALU_MFMA_BUSY_CYCLES']/1024)/(m['GRBM_GUI_ACTIVE']/8):.3f} wait_any {m['SQ_WAIT_ANY']/wc:.3f} wait_inst {m['SQ_WAIT_INST_ANY']/wc:.3f} valu {m['SQ_ACTIVE_INST_VALU']/wc:.3f} lds {m['SQ_ACTIVE_INST_LDS']/wc:.3f} vmem {m['SQ_ACTIVE_INST_VMEM']/wc:.3f} | valu/mfma insts {m.get('SQ_INSTS_VALU',0)/max(1,m.get('SQ_INSTS_MFMA',1)):.2f} lds/mfma {m.get('SQ_INSTS_LDS',0)/max(1,m.get('SQ_INSTS_MFMA',1)):.2f} bankconf/ldsactive {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):.3f}")
PY
