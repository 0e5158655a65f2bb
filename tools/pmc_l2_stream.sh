#!/bin/bash
# L2 request counters of the MLP kernels of one train step (run ON the GPU box): tools/pmc_l2_stream.sh <precision>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
i=0
for G in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcl2_$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pmcl2_$i -o p -- python $ROOT/tools/probe_step.py $P 1024 > /tmp/pmcl2_$i.log 2>&1
  F=$(find /tmp/pmcl2_$i -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { tail -3 /tmp/pmcl2_$i.log; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad")): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k[:58].ljust(58), {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
