#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 900 python tools/probe_bucket.py 1024 2048 3072 4096 > $O/probe_bucket.txt 2>&1
tail -120 $O/probe_bucket.txt
