#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --no-cpu-baseline --no-image --steps 10 --warmup 2 > $OUT/log.txt 2>&1
f=$(find $OUT/stats -name "b_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print(f"{r['Name'][:66]:66s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
