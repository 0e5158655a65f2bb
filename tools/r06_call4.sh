#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
(SCADE_BUCKET_JSON=$O/bucket_grads.json timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/gputests.txt
PRECS=f32,f16x3,bf16,bf16-s8 timeout 900 python tools/probe_bucket.py 1024 3072 4096 > $O/probe_bucket.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-image --no-rayops > $O/bench_quick.json 2> $O/bench_quick.err
tail -25 $O/gputests.txt; grep "fine.pts_linears.1\|fine.feature" $O/probe_bucket.txt
