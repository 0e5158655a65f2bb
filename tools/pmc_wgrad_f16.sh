#!/bin/bash
# SQ counters of mlp_wgrad_f16_kernel (run ON the GPU box): three rocprofv3 --pmc passes over the f16x3 train step, means per launch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmcx_$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pmcx_$i -o p -- python $ROOT/tools/probe_step.py f16x3 1024 > /tmp/pmcx_$i.log 2>&1
  F=$(find /tmp/pmcx_$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "wgrad_f16_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print({c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
