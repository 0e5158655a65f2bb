#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -k "step_finish" 2>&1 | tail -5) > $O/gputests.txt
timeout 900 bash tools/pmc_kloop_wlds.sh > $O/kloop_wlds.txt 2>&1
cat $O/gputests.txt; cat $O/kloop_wlds.txt
