#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the headline bench line un-profiled, then rocprofv3 per-kernel stats +
# separate PMC passes of the same commands ON THE SAME BOX (clock / power state differ from box to box and
# under the profiler, so only same-box figures are comparable).
# Usage: tools/profile_gpu.sh <round-tag> [render-only]   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r04}
MODE=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are only meaningful when the kernels do not overlap: keep the
# Trainer's coarse stage on the main stream while profiling (the default since round 2)
STEPS=100
# the headline region only; 12 setup + 20 warm-up steps precede the 100 timed ones (bench.py), and the
# statistics below are taken over the LAST 100 launches of each launch size, i.e. the timed steps alone
CMD="python $ROOT/bench.py --no-cpu-baseline --no-train --no-image --no-fast --no-rayops --steps $STEPS --warmup 20"
TRAIN_CMD="python $ROOT/bench.py --no-cpu-baseline --no-image --no-rayops --no-graph --steps 10 --warmup 2"   # render + f16x3 + train regions

# 0. the same command un-profiled, same box, right before the profiled runs
$CMD > $OUT/unprofiled.log 2>&1
grep -o '^{"metric.*}' $OUT/unprofiled.log | tail -1 > $OUT/bench_line_unprofiled_same_box.json

# 1. kernel trace + stats of the headline command; per-dispatch rows -> timed-launch statistics per launch size
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
grep -o '^{"metric.*}' $OUT/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
TR=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_stats.py timed "$TR" $STEPS scade > $OUT/kernel_stats_timed.csv
cat $OUT/kernel_stats_timed.csv

# 2. effective clock + MFMA busy of the headline kernel over the timed launches (its own PMC pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/clock -o bench -- $CMD > $OUT/clock.log 2>&1
grep -o '^{"metric.*}' $OUT/clock.log | tail -1 > $OUT/bench_line_under_pmc.json
CC=$(find $OUT/clock -name "*counter_collection.csv" | head -1)
CT=$(find $OUT/clock -name "*kernel_trace.csv" | head -1)
head -1 "$CC"
python $ROOT/tools/trace_stats.py clock "$CC" "$CT" $STEPS scade > $OUT/render_clock.csv
cat $OUT/render_clock.csv

if [ "$MODE" != "render-only" ]; then
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
# effective clock of every kernel of the train command (all launches of the SQ/GRBM pass)
CC=$(find $OUT/pmc3 -name "*counter_collection.csv" | head -1)
CT=$(find $OUT/pmc3 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_stats.py clock "$CC" "$CT" 100000 scade > $OUT/train_clock.csv
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
PY
python $ROOT/tools/pmc_to_json.py $OUT/pmc_summary.csv $OUT/pmc.json $OUT/train_clock.csv $OUT/render_clock.csv > $OUT/pmc_derived.txt
cat $OUT/pmc_derived.txt
cp $OUT/stats_train/*/bench_kernel_stats.csv $OUT/kernel_stats_train.csv 2>/dev/null || cp $OUT/stats_train/bench_kernel_stats.csv $OUT/kernel_stats_train.csv
fi
cp $OUT/stats/*/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
