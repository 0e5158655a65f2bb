#!/bin/bash
# tools/probe_kloop_wlds: timing of the k-loop forms, then (best effort, bounded) their L2 -> CU request counters
# (run ON the GPU box; the binary is built in the container: tools/scratch/ travels with the snapshot)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/scratch/probe_kloop_wlds
cd /tmp && export TMPDIR=/tmp
true
for M in 0 1 4; do
  rm -rf /tmp/pmck_$M
  timeout 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmck_$M -o p -- $B $M > /tmp/pmck_$M.log 2>&1
  F=$(find /tmp/pmck_$M -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "mode $M: no counters"; tail -3 /tmp/pmck_$M.log; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k[:40].ljust(40), {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
