#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O
cd $R
timeout 200 tools/timeline.sh f32_128 30 python $R/tools/probe_step.py f32 128 graph > /dev/null 2>&1
cp $R/gpurun_out/timeline_f32_128.txt $O/
(timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3) > $O/gputests_subset.txt
cat $O/gputests_subset.txt; tail -4 $O/timeline_f32_128.txt
