#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O
cd $R
timeout 700 bash tools/pmc_kloop_wlds.sh > $O/kloop_wlds.txt 2>&1
cat $O/kloop_wlds.txt | cut -c1-300
