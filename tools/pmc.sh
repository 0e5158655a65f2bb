#!/bin/bash
# Run ON THE GPU BOX: PMC counters (one group per pass, kernel-trace only) of a command, per-kernel means.
#   tools/pmc.sh <tag> <kernel-name-substring> <command...>   -> gpurun_out/pmc_<tag>.csv
TAG=$1; KSUB=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
D=/tmp/pmc_$TAG
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $D/p$i -o k -- "$@" > $D/p$i.log 2>&1
done
python3 - "$D" "$KSUB" "$ROOT/gpurun_out/pmc_$TAG.csv" <<'PY'
import csv, glob, os, sys, collections
d, ksub, out = sys.argv[1:4]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if ksub in k:
            rows[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out, "w") as fo:
    fo.write("kernel,counter,dispatches,mean,min,max\n")
    for k in sorted(rows):
        for c, v in sorted(rows[k].items()):
            line = f'"{k}",{c},{len(v)},{sum(v)/len(v):.6g},{min(v):.6g},{max(v):.6g}'
            fo.write(line + "\n"); print(line)
PY
