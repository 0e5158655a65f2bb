#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
timeout 600 python tools/probe_s8_underflow.py 1024 2048 > $O/probe_s8.txt 2>&1
tail -60 $O/probe_s8.txt
