#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 per-kernel stats of a command, top rows printed.
#   tools/kstats.sh <tag> <command...>      -> gpurun_out/kstats_<tag>.csv
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
D=/tmp/kstats_$TAG
rm -rf $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- "$@" > $D.log 2>&1
F=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1); if [ -z "$F" ]; then tail -20 $D.log; exit 1; fi
cp "$F" $ROOT/gpurun_out/kstats_$TAG.csv
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:86]:86s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}  {float(r["Percentage"]):5.1f}%')
PY
