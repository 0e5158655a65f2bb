#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > $O/gputests.txt
timeout 400 python bench.py --no-cpu-baseline --no-image --no-rayops > $O/bench_quick.json 2> $O/bench_quick.err
for p in f32 bf16-s8; do
  timeout 200 tools/timeline.sh ${p}_128 30 python $R/tools/probe_step.py $p 128 graph > /dev/null 2>&1
done
cp $R/gpurun_out/timeline_*_128.txt $O/ 2>/dev/null; true
tail -12 $O/gputests.txt; ls $R/gpurun_out | head -30
