#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_exact; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ex -- python $ROOT/tools/scratch/exact_loop.py > $OUT/log.txt 2>&1
f=$(find $OUT/stats -name "ex_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/stats
