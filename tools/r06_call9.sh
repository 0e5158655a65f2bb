#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/gputests.txt
timeout 400 python bench.py --no-cpu-baseline --no-image --no-rayops > $O/bench_quick.json 2> $O/bench_quick.err
timeout 200 tools/timeline.sh bf16-s8_128 30 python $R/tools/probe_step.py bf16-s8 128 graph > /dev/null 2>&1
cp $R/gpurun_out/timeline_bf16-s8_128.txt $O/ 2>/dev/null
tail -12 $O/gputests.txt
