#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 per-kernel stats + separate PMC passes of the
# default bench command.  Usage: tools/profile_gpu.sh <round-tag>   (outputs under gpurun_out/)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are only meaningful when the kernels do not overlap: keep the
# Trainer's coarse stage on the main stream while profiling (the default since round 2; f16x3 would
# otherwise turn its side stream on)
export SCADE_OVERLAP_COARSE=0
# the headline region only, so per-kernel averages are those of the timed render steps
CMD="python $ROOT/bench.py --no-cpu-baseline --no-train --no-image --no-fast --no-rayops --steps 20 --warmup 3"
TRAIN_CMD="python $ROOT/bench.py --no-cpu-baseline --no-image --no-rayops --steps 10 --warmup 2"   # render + f16x3 + train regions
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
grep -o '^{"metric.*}' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o bench -- $TRAIN_CMD > $OUT/stats_train.log 2>&1
grep -o '^{"metric.*}' $OUT/stats_train.log | tail -1 > $OUT/bench_line_train_under_rocprof.json
# PMC passes: one counter group per run, kernel-trace only (no other trace domains)
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d $OUT/pmc$i -o bench -- $TRAIN_CMD > $OUT/pmc$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.csv"), "w") as fo:
    fo.write("kernel,counter,dispatches,mean_per_dispatch,total\n")
    for k in sorted(rows):
        if "scade" not in k:
            continue
        for c, v in sorted(rows[k].items()):
            fo.write(f"\"{k[:90]}\",{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}\n")
print(open(os.path.join(out, "pmc_summary.csv")).read()[:6000])
PY
python $ROOT/tools/pmc_to_json.py $OUT/pmc_summary.csv $OUT/pmc.json > $OUT/pmc_derived.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cp $OUT/stats/*/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/stats_train/*/bench_kernel_stats.csv $OUT/kernel_stats_train.csv 2>/dev/null || cp $OUT/stats_train/bench_kernel_stats.csv $OUT/kernel_stats_train.csv
du -sh $OUT
